// Device-resident geometry of whole frames: network outputs -> poses without a host round trip.
//
// The reference leaves the device right after the network -- three .cpu() synchronisations (lib/object_slam.py:1100-1109) -- and does
// everything else in Python: mask logic, per-object compaction `exp_uv[k][kp_mask]` (:1122-1135), the K^-T normalisation inside pnp()
// (:34-36), one lambdatwist.pnp call per object (:1144), the acceptance test (:1147-1148), and in optimize() one g2o edge per keypoint
// with np.linalg.inv(cov) as information (:795-837).  On this part that host round trip is most of a frame at 8 crops.  Here the same
// steps are four small kernels on the caller's stream, fed by the device buffers suo_net_forward* / suo_keypoint_masks wrote:
//
//   fg_prep_kernel      wave per crop: ballot-compacts the valid keypoints (mask order = the reference's boolean indexing), widens
//                       uv / model points to fp64, normalises uv with the host-inverted K_bbox (ys, PnP input), inverts the 2x2
//                       covariances (information of the graph edges), copies uv / cov / mask into the read-back block
//   pnp_batch_kernel    csrc/pnp.hip, unchanged arithmetic; point counts read from the device (`counts`)
//   fg_build_kernel     wave per frame: acceptance (pose found, T[2,3] > 0.5 diameter, >= 4 keypoints), then the frame's pose graph as
//                       the LmProblem the LM kernels take -- fixed identity camera, one vertex per accepted crop initialised with its PnP
//                       pose, edges = the crop's compacted keypoints in their slots (41 per crop), rejected crops as fixed vertices
//                       without edges
//   lm_frame2_kernel    csrc/lm_frame2.hip: the robust rounds of ObjectSLAM.optimize (:842-896), a wave per frame
//
// and ONE device-to-host copy of everything the host keeps (PnP poses, refined poses, inlier flags, chi2, keypoints, covariances, masks,
// counts, LM statistics).  PnP inputs are bit-identical to the host route's (suo_slam_amd/object_slam.py: _run_kp_model), so the PnP
// poses are; the information matrices are the fp64 closed-form inverse of the float32 covariance where the host route rounds
// np.linalg.inv to float32 first (LAPACK's float32 arithmetic is not reproducible outside it): refined poses agree to the LM tolerance.
#include <string.h>

#include <algorithm>
#include <vector>

#include "../../include/suo_hip.h"
#include "lm_device.h"
#include "suo_internal.h"

namespace suo {

int pnp_get_iterations(double estimated_inliers);
int launch_pnp_batch_counts(int n_obj, const int* offsets, const int* counts, const int* group_first, const double* xs, const double* ys, double threshold, uint64_t seed,
                            const int* iter_tab, const int* iter_tab_off, int do_refine, double* T_out, int* status, int* best_out,
                            int* iters_out, hipStream_t s, const uint64_t* seed_add = nullptr);
int launch_lm_frame2(const void* problems_dev, int n_problems, int max_obj, int max_edges, hipStream_t s);

constexpr int FG_MAX_OBJ = 16;          // objects per frame the one-wave-per-object LM kernel takes (csrc/lm_frame.hip: LF_MAX_OBJ)

struct FgArrays {                       // device pointers into the context's arena (fixed at creation)
    // staged from the host per launch
    const int* frame_first;             // [F + 1] crop range of each frame
    const double* kinv;                 // [L][6]  KinvT[0][0], [1][0], [2][0], [0][1], [1][1], [2][1] of inv(K_bbox).T
    const double* camk;                 // [L][4]  fx, fy, cx, cy of K_bbox
    const double* min_depth;            // [L]     0.5 * diameter
    const int* crop_frame_first;        // [L]     first crop of the crop's frame
    // PnP problems, 41 slots per crop
    double* xs; double* ys; int* counts; int* offsets; int* tab_off;
    const int* iter_tab;                // get_iterations(b / n) for every n <= 41, b <= n, at n (n + 1) / 2 + b
    // graph
    double* edge_k; double* edge_uv; double* edge_info; int* edge_pair; uint8_t* level; double* err_unused;
    int* pair_start; int* pair_end; int* pair_cam; const int* iota;
    double* cam_T; uint8_t* cam_fixed; uint8_t* obj_fixed;
    LmProblem* problems;
    // read-back block
    double* T_pnp; double* T_opt; double* chi2; int* status; int* best; int* iters; int* counts_out; int* stats;
    uint8_t* accepted; uint8_t* inlier; float* uv_out; float* cov_out; uint8_t* mask_out;
};

// ---- compaction + normalisation + information (lib/object_slam.py:1118-1135, :34-36, :825-828) ----------------------------------
__global__ __launch_bounds__(64) void fg_prep_kernel(FgArrays A, const float* __restrict__ uv, const float* __restrict__ cov,
                                                     const uint8_t* __restrict__ mask, const float* __restrict__ model_kps, int use_cov) {
    const int g = blockIdx.x, k = threadIdx.x;
    const bool valid = k < NUM_KP && mask[g * NUM_KP + k] != 0;
    const unsigned long long b = __ballot(valid);
    const int pos = __popcll(b & ((1ull << k) - 1ull));
    const int n = __popcll(b);
    if (k < NUM_KP) {
        const int i = g * NUM_KP + k;
        A.uv_out[2 * i] = uv[2 * i]; A.uv_out[2 * i + 1] = uv[2 * i + 1];
        for (int t = 0; t < 4; ++t) A.cov_out[4 * i + t] = cov[4 * i + t];
        A.mask_out[i] = valid ? 1 : 0;
    }
    if (valid) {
        const int i = g * NUM_KP + k, e = g * NUM_KP + pos;
        const double u = (double)uv[2 * i], v = (double)uv[2 * i + 1];
        const double* Ki = A.kinv + 6 * g;
        for (int t = 0; t < 3; ++t) A.xs[3 * e + t] = (double)model_kps[3 * i + t];
        // points_2d @ KinvT[:2,:2] + KinvT[2:3,:2]  (two products summed, then the offset: numpy's order for a skew-free K)
        A.ys[2 * e] = (u * Ki[0] + v * Ki[1]) + Ki[2];
        A.ys[2 * e + 1] = (u * Ki[3] + v * Ki[4]) + Ki[5];
        A.edge_uv[2 * e] = u; A.edge_uv[2 * e + 1] = v;
        for (int t = 0; t < 4; ++t) A.edge_k[4 * e + t] = A.camk[4 * g + t];
        double ixx = 1.0, ixy = 0.0, iyy = 1.0;                    // Inf = np.eye(2) without network covariance (:825)
        if (use_cov) {
            const double a = (double)cov[4 * i], bb = (double)cov[4 * i + 1], c = (double)cov[4 * i + 2], d = (double)cov[4 * i + 3];
            const double det = a * d - bb * c;
            ixx = d / det; iyy = a / det; ixy = 0.5 * (-bb / det + -c / det);
        }
        A.edge_info[3 * e] = ixx; A.edge_info[3 * e + 1] = ixy; A.edge_info[3 * e + 2] = iyy;
    }
    if (k == 0) {
        A.counts[g] = n; A.counts_out[g] = n;
        A.offsets[g] = g * NUM_KP;
        A.tab_off[g] = n * (n + 1) / 2;
    }
}

// ---- acceptance (:1143-1165) + graph of the frame (:746-839) as an LmProblem -------------------------------------------------------
// seed_run: the caller's device-resident running sampler key (suo_frame_geom_params.seed_dev) -- advanced here, AFTER the PnP launch of this chain read it, by the
// frame's number of solvable problems: what a host caller adds to its seed between launches (suo_slam_amd/object_slam.py), without the read-back in between
__global__ __launch_bounds__(64) void fg_build_kernel(FgArrays A, int n_rounds, int its0, int its1, int its2, int its3, double chi2_thr,
                                                      double huber_delta, unsigned long long* seed_run) {
    const int f = blockIdx.x, lane = threadIdx.x;
    const int g0 = A.frame_first[f], nobj = A.frame_first[f + 1] - g0;
    int ne = 0, nsolv = 0;
    if (lane < nobj) {
        const int g = g0 + lane;
        const int n = A.counts[g];
        nsolv = n >= 4 ? 1 : 0;
        const double* T = A.T_pnp + 16 * g;
        const bool ok = A.status[g] == 0 && n >= 4 && T[11] > A.min_depth[g];      // T[2][3] > 0.5 * diameter
        A.accepted[g] = ok ? 1 : 0;
        A.obj_fixed[g] = ok ? 0 : 1;
        for (int t = 0; t < 12; ++t) A.T_opt[12 * g + t] = T[t];                  // rows 0..2 of the 4x4
        A.pair_start[g] = lane * NUM_KP;
        A.pair_end[g] = lane * NUM_KP + (ok ? n : 0);
        A.pair_cam[g] = 0;
        for (int j = 0; j < NUM_KP; ++j) { A.edge_pair[g * NUM_KP + j] = lane; A.level[g * NUM_KP + j] = 0; A.inlier[g * NUM_KP + j] = 1; }
        ne = ok ? n : 0;
    }
#pragma unroll
    for (int m = 32; m > 0; m >>= 1) { ne += __shfl_xor(ne, m, 64); nsolv += __shfl_xor(nsolv, m, 64); }
    if (seed_run && lane == 0 && nsolv) atomicAdd(seed_run, (unsigned long long)nsolv);
    if (lane < 12) A.cam_T[12 * f + lane] = (lane % 5 == 0) ? 1.0 : 0.0;           // T_GtoC = eye(4)[:3] (:383-385), fixed (:774)
    if (lane == 0) {
        A.cam_fixed[f] = 1;
        LmProblem P;
        memset(&P, 0, sizeof(P));
        const size_t e0 = (size_t)g0 * NUM_KP;
        P.n_cam = 1; P.n_obj = nobj; P.n_edge = ne; P.n_pair = nobj;
        P.cam_T = A.cam_T + 12 * f; P.obj_T = A.T_opt + 12 * g0;
        P.cam_fixed = A.cam_fixed + f; P.obj_fixed = A.obj_fixed + g0;
        P.edge_pair = A.edge_pair + e0;
        P.edge_k = A.edge_k + 4 * e0; P.edge_p = A.xs + 3 * e0; P.edge_uv = A.edge_uv + 2 * e0; P.edge_info = A.edge_info + 3 * e0;
        P.edge_inlier = A.inlier + e0; P.edge_chi2 = A.chi2 + e0;
        P.pair_cam = A.pair_cam + g0; P.pair_obj = A.iota; P.pair_start = A.pair_start + g0; P.pair_end = A.pair_end + g0;
        P.obj_pair_ptr = A.iota; P.obj_pair_idx = A.iota;
        P.its[0] = its0; P.its[1] = its1; P.its[2] = its2; P.its[3] = its3;
        P.n_rounds = n_rounds; P.init_with_outliers = 0; P.chi2_thr = chi2_thr; P.huber_delta = huber_delta;
        P.level = A.level + e0;
        P.stats = A.stats + 4 * f;
        A.problems[f] = P;
        for (int t = 0; t < 4; ++t) A.stats[4 * f + t] = 0;
    }
}

}  // namespace suo

using namespace suo;

struct suo_frame_geom {
    int max_crops = 0, max_frames = 0;
    char* dev = nullptr; char* host = nullptr;          // one arena each; host pinned
    size_t in_bytes = 0, out_off = 0, out_bytes = 0, total = 0;
    FgArrays A;                                         // device pointers
    FgArrays H;                                         // the same layout over the pinned host block (staged inputs + read-back block)
    hipEvent_t done = nullptr;
    int n_frames = 0, L = 0, launched = 0;
};

namespace {
struct Lay { size_t off = 0; size_t take(size_t b) { size_t o = off; off = (off + b + 63) & ~(size_t)63; return o; } };
}

extern "C" {

int suo_frame_geom_create(int max_crops, int max_frames, suo_frame_geom** out) {
    if (!out || max_crops <= 0 || max_frames <= 0) { suo_set_error("suo_frame_geom_create: bad arguments"); return SUO_ERR_ARG; }
    suo_frame_geom* c = new suo_frame_geom();
    c->max_crops = max_crops; c->max_frames = max_frames;
    const size_t L = max_crops, F = max_frames, E = L * NUM_KP;
    Lay y;
    size_t o[64]; int n = 0;
    // inputs staged per launch (one H2D)
    o[n++] = y.take(sizeof(int) * (F + 1)); o[n++] = y.take(sizeof(double) * 6 * L); o[n++] = y.take(sizeof(double) * 4 * L); o[n++] = y.take(sizeof(double) * L);
    o[n++] = y.take(sizeof(int) * L);
    c->in_bytes = y.off;
    // work arrays
    const int w0 = n;
    o[n++] = y.take(sizeof(double) * 3 * E); o[n++] = y.take(sizeof(double) * 2 * E); o[n++] = y.take(sizeof(int) * L); o[n++] = y.take(sizeof(int) * (L + 1));
    o[n++] = y.take(sizeof(int) * L);
    std::vector<int> tab;
    for (int m = 0; m <= NUM_KP; ++m)
        for (int b = 0; b <= m; ++b) tab.push_back(pnp_get_iterations(m > 0 ? b / (double)m : 0.0));
    o[n++] = y.take(sizeof(int) * tab.size());
    o[n++] = y.take(sizeof(double) * 4 * E); o[n++] = y.take(sizeof(double) * 2 * E); o[n++] = y.take(sizeof(double) * 3 * E); o[n++] = y.take(sizeof(int) * E);
    o[n++] = y.take(E);
    o[n++] = y.take(sizeof(int) * L); o[n++] = y.take(sizeof(int) * L); o[n++] = y.take(sizeof(int) * L); o[n++] = y.take(sizeof(int) * (FG_MAX_OBJ + 1));
    o[n++] = y.take(sizeof(double) * 12 * F); o[n++] = y.take(F); o[n++] = y.take(L);
    o[n++] = y.take(sizeof(LmProblem) * F);
    // read-back block (one D2H)
    c->out_off = y.off;
    const int r0 = n;
    o[n++] = y.take(sizeof(double) * 16 * L); o[n++] = y.take(sizeof(double) * 12 * L); o[n++] = y.take(sizeof(double) * E);
    o[n++] = y.take(sizeof(int) * L); o[n++] = y.take(sizeof(int) * L); o[n++] = y.take(sizeof(int) * L); o[n++] = y.take(sizeof(int) * L);
    o[n++] = y.take(sizeof(int) * 4 * F);
    o[n++] = y.take(L); o[n++] = y.take(E); o[n++] = y.take(sizeof(float) * 2 * E); o[n++] = y.take(sizeof(float) * 4 * E); o[n++] = y.take(E);
    c->out_bytes = y.off - c->out_off;
    c->total = y.off;
    if (hipMalloc((void**)&c->dev, c->total) != hipSuccess || hipHostMalloc((void**)&c->host, c->total, hipHostMallocDefault) != hipSuccess ||
        hipEventCreateWithFlags(&c->done, hipEventDisableTiming) != hipSuccess) {
        suo_set_error("suo_frame_geom_create: allocation of %zu bytes failed", c->total);
        suo_frame_geom_destroy(c);
        return SUO_ERR_HIP;
    }
    auto bind = [&](char* base, FgArrays& A) {
        int k = 0;
        A.frame_first = (const int*)(base + o[k++]); A.kinv = (const double*)(base + o[k++]); A.camk = (const double*)(base + o[k++]);
        A.min_depth = (const double*)(base + o[k++]); A.crop_frame_first = (const int*)(base + o[k++]);
        A.xs = (double*)(base + o[k++]); A.ys = (double*)(base + o[k++]); A.counts = (int*)(base + o[k++]); A.offsets = (int*)(base + o[k++]);
        A.tab_off = (int*)(base + o[k++]); A.iter_tab = (const int*)(base + o[k++]);
        A.edge_k = (double*)(base + o[k++]); A.edge_uv = (double*)(base + o[k++]); A.edge_info = (double*)(base + o[k++]);
        A.edge_pair = (int*)(base + o[k++]); A.level = (uint8_t*)(base + o[k++]); A.err_unused = nullptr;
        A.pair_start = (int*)(base + o[k++]); A.pair_end = (int*)(base + o[k++]); A.pair_cam = (int*)(base + o[k++]); A.iota = (const int*)(base + o[k++]);
        A.cam_T = (double*)(base + o[k++]); A.cam_fixed = (uint8_t*)(base + o[k++]); A.obj_fixed = (uint8_t*)(base + o[k++]);
        A.problems = (LmProblem*)(base + o[k++]);
        A.T_pnp = (double*)(base + o[k++]); A.T_opt = (double*)(base + o[k++]); A.chi2 = (double*)(base + o[k++]);
        A.status = (int*)(base + o[k++]); A.best = (int*)(base + o[k++]); A.iters = (int*)(base + o[k++]); A.counts_out = (int*)(base + o[k++]);
        A.stats = (int*)(base + o[k++]);
        A.accepted = (uint8_t*)(base + o[k++]); A.inlier = (uint8_t*)(base + o[k++]); A.uv_out = (float*)(base + o[k++]);
        A.cov_out = (float*)(base + o[k++]); A.mask_out = (uint8_t*)(base + o[k++]);
        return k;
    };
    const int used = bind(c->dev, c->A);
    bind(c->host, c->H);
    (void)w0; (void)r0;
    if (used != n) { suo_set_error("suo_frame_geom_create: layout mismatch"); suo_frame_geom_destroy(c); return SUO_ERR_ARG; }
    // constants: the iteration table and 0..16
    memset(c->host, 0, c->total);
    memcpy((void*)c->H.iter_tab, tab.data(), sizeof(int) * tab.size());
    for (int i = 0; i <= FG_MAX_OBJ; ++i) ((int*)c->H.iota)[i] = i;
    if (hipMemcpy(c->dev, c->host, c->total, hipMemcpyHostToDevice) != hipSuccess) {
        suo_set_error("suo_frame_geom_create: upload failed"); suo_frame_geom_destroy(c); return SUO_ERR_HIP;
    }
    *out = c;
    return SUO_OK;
}

void suo_frame_geom_destroy(suo_frame_geom* c) {
    if (!c) return;
    if (c->done) (void)hipEventDestroy(c->done);
    if (c->dev) (void)hipFree(c->dev);
    if (c->host) (void)hipHostFree(c->host);
    delete c;
}

int suo_frame_geom_launch(suo_frame_geom* c, int n_frames, const int* frame_first, const float* uv_dev, const float* cov_dev,
                          const uint8_t* mask_dev, const float* model_kps_dev, const double* kinv, const double* camk, const double* min_depth,
                          const suo_frame_geom_params* p, void* stream) {
    if (!c || !frame_first || !uv_dev || !cov_dev || !mask_dev || !model_kps_dev || !kinv || !camk || !min_depth || !p) {
        suo_set_error("suo_frame_geom_launch: null argument"); return SUO_ERR_ARG;
    }
    if (n_frames <= 0 || n_frames > c->max_frames) { suo_set_error("suo_frame_geom_launch: %d frames (context holds %d)", n_frames, c->max_frames); return SUO_ERR_ARG; }
    const int L = frame_first[n_frames];
    int max_obj = 0;
    for (int f = 0; f < n_frames; ++f) {
        const int m = frame_first[f + 1] - frame_first[f];
        if (frame_first[0] != 0 || m < 0) { suo_set_error("suo_frame_geom_launch: frame_first must start at 0 and not decrease"); return SUO_ERR_ARG; }
        max_obj = std::max(max_obj, m);
    }
    if (L <= 0 || L > c->max_crops) { suo_set_error("suo_frame_geom_launch: %d crops (context holds %d)", L, c->max_crops); return SUO_ERR_ARG; }
    // fg_build_kernel is one wave per frame, a lane per crop: beyond 64 crops of one frame nothing would be written
    if (max_obj > 64) { suo_set_error("suo_frame_geom_launch: %d objects in one frame (the chain takes 64 without LM, %d with)", max_obj, FG_MAX_OBJ); return SUO_ERR_ARG; }
    if (p->do_lm && max_obj > FG_MAX_OBJ) { suo_set_error("suo_frame_geom_launch: %d objects in one frame (the frame LM kernel takes %d)", max_obj, FG_MAX_OBJ); return SUO_ERR_ARG; }
    if (p->n_rounds < 0 || p->n_rounds > 4) { suo_set_error("suo_frame_geom_launch: n_rounds %d", p->n_rounds); return SUO_ERR_ARG; }
    hipStream_t s = (hipStream_t)stream;
    if (c->launched) SUO_HIP_CHECK(hipEventSynchronize(c->done));      // the pinned block of the previous launch may still be in flight
    memcpy((void*)c->H.frame_first, frame_first, sizeof(int) * (n_frames + 1));
    memcpy((void*)c->H.kinv, kinv, sizeof(double) * 6 * L);
    memcpy((void*)c->H.camk, camk, sizeof(double) * 4 * L);
    memcpy((void*)c->H.min_depth, min_depth, sizeof(double) * L);
    for (int f = 0; f < n_frames; ++f)
        for (int g = frame_first[f]; g < frame_first[f + 1]; ++g) ((int*)c->H.crop_frame_first)[g] = frame_first[f];
    SUO_HIP_CHECK(hipMemcpyAsync(c->dev, c->host, c->in_bytes, hipMemcpyHostToDevice, s));
    hipLaunchKernelGGL(fg_prep_kernel, dim3(L), dim3(64), 0, s, c->A, uv_dev, cov_dev, mask_dev, model_kps_dev, p->use_cov);
    SUO_HIP_CHECK(hipGetLastError());
    int rc = launch_pnp_batch_counts(L, c->A.offsets, c->A.counts, c->A.crop_frame_first, c->A.xs, c->A.ys, p->pnp_threshold, p->seed, c->A.iter_tab, c->A.tab_off, 1,
                                     c->A.T_pnp, c->A.status, c->A.best, c->A.iters, s, p->seed_dev);
    if (rc != SUO_OK) return rc;
    const int nr = p->do_lm ? p->n_rounds : 0;
    hipLaunchKernelGGL(fg_build_kernel, dim3(n_frames), dim3(64), 0, s, c->A, nr, p->its[0], p->its[1], p->its[2], p->its[3], p->chi2_thr,
                       p->huber_delta, (unsigned long long*)p->seed_dev);
    SUO_HIP_CHECK(hipGetLastError());
    if (p->do_lm) {
        rc = launch_lm_frame2(c->A.problems, n_frames, std::max(max_obj, 1), std::max(max_obj, 1) * NUM_KP, s);
        if (rc != SUO_OK) return rc;
    }
    SUO_HIP_CHECK(hipMemcpyAsync(c->host + c->out_off, c->dev + c->out_off, c->out_bytes, hipMemcpyDeviceToHost, s));
    SUO_HIP_CHECK(hipEventRecord(c->done, s));
    c->n_frames = n_frames; c->L = L; c->launched = 1;
    return SUO_OK;
}

int suo_frame_geom_fetch(suo_frame_geom* c, suo_frame_geom_result* r) {
    if (!c || !r) { suo_set_error("suo_frame_geom_fetch: null argument"); return SUO_ERR_ARG; }
    if (!c->launched) { suo_set_error("suo_frame_geom_fetch: nothing launched"); return SUO_ERR_ARG; }
    SUO_HIP_CHECK(hipEventSynchronize(c->done));
    r->n_frames = c->n_frames; r->n_crops = c->L;
    r->T_pnp = c->H.T_pnp; r->T_opt = c->H.T_opt; r->chi2 = c->H.chi2; r->pnp_status = c->H.status; r->pnp_best_inliers = c->H.best;
    r->pnp_iterations = c->H.iters; r->n_kp = c->H.counts_out; r->lm_stats = c->H.stats; r->accepted = c->H.accepted; r->inlier = c->H.inlier;
    r->uv = c->H.uv_out; r->cov = c->H.cov_out; r->mask = c->H.mask_out;
    return SUO_OK;
}

// The same block where it lies on the DEVICE after the launch's kernels (stream-ordered: valid for work enqueued on the launch's stream behind it, until the
// context's next launch): what a kernel that continues the chain reads -- csrc/slam_vote.hip takes T_pnp / accepted / n_kp of a SLAM view's first pass from here.
int suo_frame_geom_device_result(suo_frame_geom* c, suo_frame_geom_result* r) {
    if (!c || !r) { suo_set_error("suo_frame_geom_device_result: null argument"); return SUO_ERR_ARG; }
    if (!c->launched) { suo_set_error("suo_frame_geom_device_result: nothing launched"); return SUO_ERR_ARG; }
    r->n_frames = c->n_frames; r->n_crops = c->L;
    r->T_pnp = c->A.T_pnp; r->T_opt = c->A.T_opt; r->chi2 = c->A.chi2; r->pnp_status = c->A.status; r->pnp_best_inliers = c->A.best;
    r->pnp_iterations = c->A.iters; r->n_kp = c->A.counts_out; r->lm_stats = c->A.stats; r->accepted = c->A.accepted; r->inlier = c->A.inlier;
    r->uv = c->A.uv_out; r->cov = c->A.cov_out; r->mask = c->A.mask_out;
    return SUO_OK;
}

int suo_frame_geom_ready(suo_frame_geom* c) {
    if (!c || !c->launched) return 1;
    return hipEventQuery(c->done) == hipSuccess ? 1 : 0;
}

}  // extern "C"
