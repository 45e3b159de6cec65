// Host-buffer entry points for the geometry kernels (what the reference's FFI would bind):
//   suo_pnp / suo_pnp_batch        <- lambdatwist.pnp      (thirdparty/lambdatwist/pnp_python_binding.cpp:57-62)
//   suo_optimize / suo_optimize_batch <- the g2o calls of ObjectSLAM.optimize (lib/object_slam.py:703-903)
// They stage the caller's host arrays into one device arena (one H2D), launch, and copy results back
// (one D2H).  Everything between the two copies runs on the GPU.
#include <math.h>
#include <string.h>

#include <algorithm>
#include <mutex>
#include <vector>

#include "../../include/suo_hip.h"
#include "suo_internal.h"
#include "tune.h"

namespace suo {
int launch_pnp_replay(int n_obj, const int* offsets, const double* xs, const double* ys, double threshold, const int* iter_tab, const int* iter_tab_off,
                      int do_refine, const int* draws, int n_draws, double* T_out, int* status, int* best_out, int* iters_out, int* win_out, hipStream_t s);
int launch_pnp_batch(int n_obj, const int* offsets, const double* xs, const double* ys, double threshold, uint64_t seed,
                     const int* iter_tab, const int* iter_tab_off, int do_refine, double* T_out, int* status, int* best_out,
                     int* iters_out, hipStream_t s);
int launch_lm(const void* problems_dev, int n_problems, int lds_bytes, hipStream_t s);
int launch_lm_big(const void* problems_dev, int n_problems, int lds_bytes, hipStream_t s);
#ifdef SUO_TUNING
int launch_lm_grid(const void* problem_dev, void* scratch_dev, int n_wgs, hipStream_t s);       // (csrc/lm_grid.hip: rounds 4-5's route for one large graph; tuning builds only)
#endif
int launch_lm_cam(const void* problems_dev, int n_problems, hipStream_t s);
int launch_lm_cam2(const void* problems_dev, int n_problems, int max_edges, hipStream_t s);
int lm_cam2_max_edges();
int launch_lm_frame(const void* problems_dev, int n_problems, int max_obj, hipStream_t s);
int launch_lm_frame2(const void* problems_dev, int n_problems, int max_obj, int max_edges, hipStream_t s);
int lm_frame2_max_edges();
#ifdef SUO_TUNING
size_t lm_grid_scratch_bytes();
#endif
int lm_lds_bytes(int C, int O, int E, int NP, int n_free_obj_schur);
int launch_debug_cholesky(const double* A, const double* b, int ns, double* x, int* ok, hipStream_t s);
size_t lm_problem_struct_size();
int launch_ba_init(const void* P, hipStream_t s);
size_t ba_scratch_doubles();
int launch_ba_classify(const void* P, int keep_all, double* out, double* scratch, hipStream_t s);
int launch_ba_linearize(const void* P, int robust_on, double* out, double* scratch, int rank, int world, hipStream_t s, const double* ctl = nullptr, double* copy_to = nullptr,
                        int copy_n = 0, double* fold_ctl = nullptr);
int launch_ba_schur(const void* P, double lambda, int ns, double* out, double* scratch, hipStream_t s, const double* ctl = nullptr);
int launch_ba_solve_update(const void* P, double lambda, int ns, int robust_on, const double* HB, const double* St, int expect_ok, double* out,
                           double* scratch, double* big, hipStream_t s, const double* ctl = nullptr, double* fold_ctl = nullptr);
int launch_ba_ctl_begin(double* ctl, int its, int world, hipStream_t s);
int launch_ba_ctl_lin(const void* P, double* ctl, const double* lin, double* scratch, hipStream_t s);
int launch_ba_ctl_decide(const void* P, double* ctl, const double* red, hipStream_t s);
int launch_ba_copy(const double* src, double* dst, int n, hipStream_t s);
int launch_ba_restore(const void* P, hipStream_t s);
int launch_ba_finalize(const void* P, hipStream_t s);

// PnpParams::get_iterations (thirdparty/lambdatwist/parameters.h:76-102), evaluated on the host so the
// kernel's adaptive iteration count uses the same libm as the reference would.
int pnp_get_iterations(double estimated_inliers) {
    const double p_meets = 0.9, min_probability = 0.99999;
    const unsigned max_iterations = 1000, min_iterations = 100;
    double p_inlier = std::min(0.9, estimated_inliers * p_meets);
    p_inlier = std::min(std::max(p_inlier, 1e-2), 1 - 1e-8);
    if (p_inlier < 0.01) return (int)max_iterations;
    const double p_failure = std::min(std::max(1.0 - min_probability, 1e-8), 0.01);
    const double p_good = pow(p_inlier, 4);
    const double iterations = ceil(log(p_failure) / log(1.0 - p_good)) + 50;
    if (iterations < min_iterations) return (int)min_iterations;
    if (iterations > max_iterations) return (int)max_iterations;
    return (int)iterations;
}

// grow-only device arena + pinned host mirror, one per process (guarded: the reference is single-threaded)
struct Arena {
    char* dev = nullptr; char* host = nullptr; size_t cap = 0;
    hipStream_t stream = nullptr;
    std::mutex mu;
    int ensure(size_t bytes) {
        if (!stream) {
            // highest priority: the geometry kernels are tiny (8 waves / 1 workgroup) and latency-bound; on its own
            // priority level the stream also gets its own hardware queue instead of sharing one with the CNN's streams.
            int lo = 0, hi = 0;
            SUO_HIP_CHECK(hipDeviceGetStreamPriorityRange(&lo, &hi));
            static const int prio = (int)SUO_TUNE("SUO_GEOM_PRIO", 2);      // 2: highest, 1: default, 0: lowest (A/B only)
            SUO_HIP_CHECK(hipStreamCreateWithPriority(&stream, hipStreamNonBlocking, prio == 2 ? hi : (prio == 0 ? lo : 0)));
        }
        if (bytes <= cap) return SUO_OK;
        size_t ncap = std::max(bytes, cap * 2);
        ncap = (ncap + 4095) & ~(size_t)4095;
        if (dev) (void)hipFree(dev);
        if (host) (void)hipHostFree(host);
        dev = nullptr; host = nullptr; cap = 0;
        SUO_HIP_CHECK(hipMalloc((void**)&dev, ncap));
        SUO_HIP_CHECK(hipHostMalloc((void**)&host, ncap, hipHostMallocDefault));
        cap = ncap;
        return SUO_OK;
    }
};
static Arena g_arena;

struct Layout {
    size_t off = 0;
    size_t take(size_t bytes) { size_t o = off; off = (off + bytes + 15) & ~(size_t)15; return o; }
};

// mirror of the device-side LmProblem (csrc/lm.hip) -- keep field order identical
struct LmProblemHost {
    int n_cam, n_obj, n_edge, n_pair;
    double* cam_T; double* obj_T;
    const uint8_t* cam_fixed; const uint8_t* obj_fixed;
    const int* edge_pair;
    const double* edge_k; const double* edge_p; const double* edge_uv; const double* edge_info;
    uint8_t* edge_inlier; double* edge_chi2;
    const int* pair_cam; const int* pair_obj; const int* pair_start;
    const int* cam_pair_ptr; const int* cam_pair_idx;
    const int* obj_pair_ptr; const int* obj_pair_idx;
    const int* cam_obj_pair;
    int its[8]; int n_rounds; int init_with_outliers; double chi2_thr; double huber_delta;
    void* cam; void* obj; void* cam_bak; void* obj_bak;
    double* err; double* jac; uint8_t* level; double* pair_part;
    double* Hcc; double* bc; double* Hoo; double* bo; double* Hcc_inv; double* Y; double* yc; double* xc; double* xo;
    int* obj_slot; int* stats;
    const int* pair_end;
};

struct Prep {
    std::vector<int> order, edge_pair, pair_cam, pair_obj, pair_start, cam_ptr, cam_idx, obj_ptr, obj_idx, cam_obj;
    size_t o[40];
};
struct Staged {
    std::vector<Prep> prep;
    size_t o_structs = 0, in_end = 0, out_end = 0;
    int lds_need = 0;
};

}  // namespace suo

using namespace suo;

extern "C" {

static int pnp_batch_impl(int n_obj, const int* n_pts, const double* xs, const double* ys, double threshold, uint64_t seed, const int* draws, int n_draws,
                          int do_refine, double* T_out, int* status, int* best_inliers, int* iterations, int* winner) {
    if (n_obj <= 0) return SUO_OK;
    if (!n_pts || !xs || !ys || !T_out) { suo_set_error("suo_pnp_batch: null argument"); return SUO_ERR_ARG; }
    if (draws && n_draws < pnp_get_iterations(0.0)) {
        suo_set_error("suo_pnp_replay: %d draws per object, the RANSAC loop may run %d iterations", n_draws, pnp_get_iterations(0.0));
        return SUO_ERR_ARG;
    }
    std::lock_guard<std::mutex> lock(g_arena.mu);
    std::vector<int> offsets(n_obj + 1, 0), tab_off(n_obj, 0);
    std::vector<int> tab;
    for (int o = 0; o < n_obj; ++o) {
        if (n_pts[o] < 0) { suo_set_error("suo_pnp_batch: negative point count"); return SUO_ERR_ARG; }
        offsets[o + 1] = offsets[o] + n_pts[o];
        tab_off[o] = (int)tab.size();
        for (int b = 0; b <= n_pts[o]; ++b) tab.push_back(pnp_get_iterations(n_pts[o] > 0 ? b / (double)n_pts[o] : 0.0));
    }
    const int total = offsets[n_obj];
    Layout L;
    const size_t o_off = L.take(sizeof(int) * (n_obj + 1)), o_toff = L.take(sizeof(int) * n_obj), o_tab = L.take(sizeof(int) * tab.size());
    const size_t o_xs = L.take(sizeof(double) * 3 * (size_t)total), o_ys = L.take(sizeof(double) * 2 * (size_t)total);
    const size_t o_dr = L.take(draws ? sizeof(int) * 4 * (size_t)n_draws * n_obj : 0);
    const size_t in_bytes = L.off;
    const size_t o_T = L.take(sizeof(double) * 16 * (size_t)n_obj), o_st = L.take(sizeof(int) * n_obj), o_best = L.take(sizeof(int) * n_obj),
                 o_it = L.take(sizeof(int) * n_obj), o_win = L.take(sizeof(int) * n_obj);
    int rc = g_arena.ensure(L.off);
    if (rc != SUO_OK) return rc;
    char* h = g_arena.host;
    char* d = g_arena.dev;
    memcpy(h + o_off, offsets.data(), sizeof(int) * (n_obj + 1));
    memcpy(h + o_toff, tab_off.data(), sizeof(int) * n_obj);
    memcpy(h + o_tab, tab.data(), sizeof(int) * tab.size());
    memcpy(h + o_xs, xs, sizeof(double) * 3 * (size_t)total);
    memcpy(h + o_ys, ys, sizeof(double) * 2 * (size_t)total);
    if (draws) memcpy(h + o_dr, draws, sizeof(int) * 4 * (size_t)n_draws * n_obj);
    hipStream_t s = g_arena.stream;
    SUO_HIP_CHECK(hipMemcpyAsync(d, h, in_bytes, hipMemcpyHostToDevice, s));
    if (draws)
        rc = launch_pnp_replay(n_obj, (const int*)(d + o_off), (const double*)(d + o_xs), (const double*)(d + o_ys), threshold, (const int*)(d + o_tab),
                               (const int*)(d + o_toff), do_refine, (const int*)(d + o_dr), n_draws, (double*)(d + o_T), (int*)(d + o_st), (int*)(d + o_best),
                               (int*)(d + o_it), (int*)(d + o_win), s);
    else
    rc = launch_pnp_batch(n_obj, (const int*)(d + o_off), (const double*)(d + o_xs), (const double*)(d + o_ys), threshold, seed,
                          (const int*)(d + o_tab), (const int*)(d + o_toff), do_refine, (double*)(d + o_T), (int*)(d + o_st),
                          (int*)(d + o_best), (int*)(d + o_it), s);
    if (rc != SUO_OK) return rc;
    SUO_HIP_CHECK(hipMemcpyAsync(h + o_T, d + o_T, L.off - o_T, hipMemcpyDeviceToHost, s));
    SUO_HIP_CHECK(hipStreamSynchronize(s));
    memcpy(T_out, h + o_T, sizeof(double) * 16 * (size_t)n_obj);
    if (status) memcpy(status, h + o_st, sizeof(int) * n_obj);
    if (best_inliers) memcpy(best_inliers, h + o_best, sizeof(int) * n_obj);
    if (iterations) memcpy(iterations, h + o_it, sizeof(int) * n_obj);
    if (winner && draws) memcpy(winner, h + o_win, sizeof(int) * n_obj);
    return SUO_OK;
}

int suo_pnp_batch(int n_obj, const int* n_pts, const double* xs, const double* ys, double threshold, uint64_t seed,
                  int do_refine, double* T_out, int* status, int* best_inliers, int* iterations) {
    return pnp_batch_impl(n_obj, n_pts, xs, ys, threshold, seed, nullptr, 0, do_refine, T_out, status, best_inliers, iterations, nullptr);
}

int suo_pnp_replay(int n_obj, const int* n_pts, const double* xs, const double* ys, double threshold, const int* draws, int n_draws, int do_refine,
                   double* T_out, int* status, int* best_inliers, int* iterations, int* winner) {
    if (!draws) { suo_set_error("suo_pnp_replay: draws is NULL"); return SUO_ERR_ARG; }
    return pnp_batch_impl(n_obj, n_pts, xs, ys, threshold, 0, draws, n_draws, do_refine, T_out, status, best_inliers, iterations, winner);
}

int suo_pnp(const double* xs, const double* ys, int n, double threshold, double* T_out) {
    int st = 0;
    return suo_pnp_batch(1, &n, xs, ys, threshold, 0, 1, T_out, &st, nullptr, nullptr);
}

// host prep (sort edges by (camera, object) pair, CSR) + arena layout + H2D of one batch of problems
static int stage_problems(suo_ba_problem* probs, int n_prob, Arena& A, Staged& st) {
    st.prep.assign(n_prob, Prep());
    std::vector<Prep>& prep = st.prep;
    Layout L;
    const size_t o_structs = L.take(sizeof(LmProblemHost) * (size_t)n_prob);
    st.o_structs = o_structs;
    if (sizeof(LmProblemHost) != lm_problem_struct_size()) { suo_set_error("LmProblem layout mismatch"); return SUO_ERR_ARG; }
    size_t in_end = 0;
    for (int i = 0; i < n_prob; ++i) {
        suo_ba_problem& q = probs[i];
        Prep& P = prep[i];
        if (q.n_cam < 0 || q.n_obj < 0 || q.n_edge < 0 || q.n_rounds < 0 || q.n_rounds > 8) { suo_set_error("suo_optimize: bad sizes"); return SUO_ERR_ARG; }
        // sort edges by (cam, obj) pair, stable, so each pair is one contiguous segment
        P.order.resize(q.n_edge);
        for (int e = 0; e < q.n_edge; ++e) {
            if (q.edge_cam[e] < 0 || q.edge_cam[e] >= q.n_cam || q.edge_obj[e] < 0 || q.edge_obj[e] >= q.n_obj) {
                suo_set_error("suo_optimize: edge %d references a missing vertex", e);
                return SUO_ERR_ARG;
            }
            P.order[e] = e;
        }
        std::stable_sort(P.order.begin(), P.order.end(), [&](int a, int b) {
            if (q.edge_cam[a] != q.edge_cam[b]) return q.edge_cam[a] < q.edge_cam[b];
            return q.edge_obj[a] < q.edge_obj[b];
        });
        P.edge_pair.resize(q.n_edge);
        for (int k = 0; k < q.n_edge; ++k) {
            const int e = P.order[k];
            if (k == 0 || q.edge_cam[e] != q.edge_cam[P.order[k - 1]] || q.edge_obj[e] != q.edge_obj[P.order[k - 1]]) {
                P.pair_cam.push_back(q.edge_cam[e]);
                P.pair_obj.push_back(q.edge_obj[e]);
                P.pair_start.push_back(k);
            }
            P.edge_pair[k] = (int)P.pair_cam.size() - 1;
        }
        P.pair_start.push_back(q.n_edge);
        const int np = (int)P.pair_cam.size();
        P.cam_ptr.assign(q.n_cam + 1, 0);
        P.obj_ptr.assign(q.n_obj + 1, 0);
        for (int p = 0; p < np; ++p) { P.cam_ptr[P.pair_cam[p] + 1]++; P.obj_ptr[P.pair_obj[p] + 1]++; }
        for (int c = 0; c < q.n_cam; ++c) P.cam_ptr[c + 1] += P.cam_ptr[c];
        for (int o = 0; o < q.n_obj; ++o) P.obj_ptr[o + 1] += P.obj_ptr[o];
        P.cam_idx.resize(np);
        P.obj_idx.resize(np);
        { std::vector<int> cc(P.cam_ptr.begin(), P.cam_ptr.end() - 1), oo(P.obj_ptr.begin(), P.obj_ptr.end() - 1);
          for (int p = 0; p < np; ++p) { P.cam_idx[cc[P.pair_cam[p]]++] = p; P.obj_idx[oo[P.pair_obj[p]]++] = p; } }
        P.cam_obj.assign((size_t)q.n_cam * q.n_obj, -1);                    // dense (camera, object) -> pair lookup for the Schur sums
        for (int p = 0; p < np; ++p) P.cam_obj[(size_t)P.pair_cam[p] * q.n_obj + P.pair_obj[p]] = p;
        const size_t E = q.n_edge, C = q.n_cam, O = q.n_obj, NP = np;
        size_t* o = P.o;
        o[0] = L.take(sizeof(double) * 12 * C); o[1] = L.take(sizeof(double) * 12 * O);           // cam_T, obj_T
        o[2] = L.take(C); o[3] = L.take(O);                                                      // fixed flags
        o[4] = L.take(sizeof(int) * E);                                                          // edge_pair
        o[5] = L.take(sizeof(double) * 4 * E); o[6] = L.take(sizeof(double) * 3 * E);
        o[7] = L.take(sizeof(double) * 2 * E); o[8] = L.take(sizeof(double) * 3 * E);
        o[9] = L.take(E);                                                                        // edge_inlier
        o[10] = L.take(sizeof(int) * NP); o[11] = L.take(sizeof(int) * NP); o[12] = L.take(sizeof(int) * (NP + 1));
        o[13] = L.take(sizeof(int) * (C + 1)); o[14] = L.take(sizeof(int) * NP);
        o[15] = L.take(sizeof(int) * (O + 1)); o[16] = L.take(sizeof(int) * NP);
        o[37] = L.take(sizeof(int) * C * O);                                                     // cam_obj_pair
    }
    in_end = L.off;
    // outputs + scratch (device only, but laid out in the same arena; outputs first for one D2H)
    for (int i = 0; i < n_prob; ++i) {
        suo_ba_problem& q = probs[i];
        size_t* o = prep[i].o;
        o[17] = L.take(sizeof(double) * q.n_edge);     // edge_chi2
        o[18] = L.take(sizeof(int) * 4);               // stats
    }
    const size_t out_end = L.off;
    for (int i = 0; i < n_prob; ++i) {
        suo_ba_problem& q = probs[i];
        size_t* o = prep[i].o;
        const size_t E = q.n_edge, C = q.n_cam, O = q.n_obj, NP = prep[i].pair_cam.size();
        o[19] = L.take(56 * C); o[20] = L.take(56 * O); o[21] = L.take(56 * C); o[22] = L.take(56 * O);   // Pose = 7 doubles
        o[23] = L.take(sizeof(double) * 2 * E); o[24] = L.take(E); o[25] = L.take(sizeof(double) * 90 * NP);
        o[26] = L.take(sizeof(double) * 36 * C); o[27] = L.take(sizeof(double) * 6 * C);
        o[28] = L.take(sizeof(double) * 36 * O); o[29] = L.take(sizeof(double) * 6 * O);
        o[30] = L.take(sizeof(double) * 36 * C); o[31] = L.take(sizeof(double) * 36 * NP); o[32] = L.take(sizeof(double) * 6 * C);
        o[33] = L.take(sizeof(double) * 6 * C); o[34] = L.take(sizeof(double) * 6 * O); o[35] = L.take(sizeof(int) * O);
        o[36] = L.take(sizeof(double) * 29 * E);
    }
    st.in_end = in_end;
    st.out_end = out_end;
    int rc = A.ensure(L.off);
    if (rc != SUO_OK) return rc;
    char* h = A.host;
    char* d = A.dev;
    LmProblemHost* hs = (LmProblemHost*)(h + o_structs);
    for (int i = 0; i < n_prob; ++i) {
        suo_ba_problem& q = probs[i];
        Prep& P = prep[i];
        size_t* o = P.o;
        const int E = q.n_edge, np = (int)P.pair_cam.size();
        memcpy(h + o[0], q.cam_T, sizeof(double) * 12 * q.n_cam);
        memcpy(h + o[1], q.obj_T, sizeof(double) * 12 * q.n_obj);
        memcpy(h + o[2], q.cam_fixed, q.n_cam);
        memcpy(h + o[3], q.obj_fixed, q.n_obj);
        memcpy(h + o[4], P.edge_pair.data(), sizeof(int) * E);
        for (int k = 0; k < E; ++k) {
            const int e = P.order[k];
            memcpy(h + o[5] + sizeof(double) * 4 * k, q.edge_camk + 4 * e, sizeof(double) * 4);
            memcpy(h + o[6] + sizeof(double) * 3 * k, q.edge_p + 3 * e, sizeof(double) * 3);
            memcpy(h + o[7] + sizeof(double) * 2 * k, q.edge_uv + 2 * e, sizeof(double) * 2);
            memcpy(h + o[8] + sizeof(double) * 3 * k, q.edge_info + 3 * e, sizeof(double) * 3);
            ((uint8_t*)(h + o[9]))[k] = q.edge_inlier[e];
        }
        memcpy(h + o[10], P.pair_cam.data(), sizeof(int) * np);
        memcpy(h + o[11], P.pair_obj.data(), sizeof(int) * np);
        memcpy(h + o[12], P.pair_start.data(), sizeof(int) * (np + 1));
        memcpy(h + o[13], P.cam_ptr.data(), sizeof(int) * (q.n_cam + 1));
        memcpy(h + o[14], P.cam_idx.data(), sizeof(int) * np);
        memcpy(h + o[15], P.obj_ptr.data(), sizeof(int) * (q.n_obj + 1));
        memcpy(h + o[16], P.obj_idx.data(), sizeof(int) * np);
        memcpy(h + o[37], P.cam_obj.data(), sizeof(int) * P.cam_obj.size());
        LmProblemHost& S = hs[i];
        memset(&S, 0, sizeof(S));
        S.n_cam = q.n_cam; S.n_obj = q.n_obj; S.n_edge = E; S.n_pair = np;
        S.cam_T = (double*)(d + o[0]); S.obj_T = (double*)(d + o[1]);
        S.cam_fixed = (const uint8_t*)(d + o[2]); S.obj_fixed = (const uint8_t*)(d + o[3]);
        S.edge_pair = (const int*)(d + o[4]);
        S.edge_k = (const double*)(d + o[5]); S.edge_p = (const double*)(d + o[6]); S.edge_uv = (const double*)(d + o[7]);
        S.edge_info = (const double*)(d + o[8]); S.edge_inlier = (uint8_t*)(d + o[9]); S.edge_chi2 = (double*)(d + o[17]);
        S.pair_cam = (const int*)(d + o[10]); S.pair_obj = (const int*)(d + o[11]); S.pair_start = (const int*)(d + o[12]);
        S.cam_pair_ptr = (const int*)(d + o[13]); S.cam_pair_idx = (const int*)(d + o[14]);
        S.obj_pair_ptr = (const int*)(d + o[15]); S.obj_pair_idx = (const int*)(d + o[16]);
        S.cam_obj_pair = (const int*)(d + o[37]);
        for (int k = 0; k < 8; ++k) S.its[k] = k < q.n_rounds ? q.its[k] : 0;
        S.n_rounds = q.n_rounds; S.init_with_outliers = q.init_with_outliers; S.chi2_thr = q.chi2_thr; S.huber_delta = q.huber_delta;
        S.cam = d + o[19]; S.obj = d + o[20]; S.cam_bak = d + o[21]; S.obj_bak = d + o[22];
        S.jac = (double*)(d + o[36]);
        S.err = (double*)(d + o[23]); S.level = (uint8_t*)(d + o[24]); S.pair_part = (double*)(d + o[25]);
        S.Hcc = (double*)(d + o[26]); S.bc = (double*)(d + o[27]); S.Hoo = (double*)(d + o[28]); S.bo = (double*)(d + o[29]);
        S.Hcc_inv = (double*)(d + o[30]); S.Y = (double*)(d + o[31]); S.yc = (double*)(d + o[32]);
        S.xc = (double*)(d + o[33]); S.xo = (double*)(d + o[34]); S.obj_slot = (int*)(d + o[35]); S.stats = (int*)(d + o[18]);
    }
    SUO_HIP_CHECK(hipMemcpyAsync(d, h, in_end, hipMemcpyHostToDevice, A.stream));
    st.lds_need = 0;
    for (int i = 0; i < n_prob; ++i) {
        const suo_ba_problem& q = probs[i];
        int nfo = 0, nfc = 0;
        for (int o = 0; o < q.n_obj; ++o) nfo += q.obj_fixed[o] ? 0 : 1;
        for (int c = 0; c < q.n_cam; ++c) nfc += q.cam_fixed[c] ? 0 : 1;
        st.lds_need = std::max(st.lds_need, lm_lds_bytes(q.n_cam, q.n_obj, q.n_edge, (int)prep[i].pair_cam.size(), (nfo > 0 && nfc > 0) ? nfo : 0));
    }
    return SUO_OK;
}

// D2H of poses / inlier flags / chi2 / stats and un-sorting into the caller's arrays
static int fetch_results(suo_ba_problem* probs, int n_prob, Arena& A, Staged& st) {
    char* h = A.host;
    char* d = A.dev;
    std::vector<Prep>& prep = st.prep;
    SUO_HIP_CHECK(hipMemcpyAsync(h, d, st.out_end, hipMemcpyDeviceToHost, A.stream));
    SUO_HIP_CHECK(hipStreamSynchronize(A.stream));
    for (int i = 0; i < n_prob; ++i) {
        suo_ba_problem& q = probs[i];
        Prep& P = prep[i];
        size_t* o = P.o;
        memcpy(q.cam_T, h + o[0], sizeof(double) * 12 * q.n_cam);
        memcpy(q.obj_T, h + o[1], sizeof(double) * 12 * q.n_obj);
        for (int k = 0; k < q.n_edge; ++k) {
            const int e = P.order[k];
            q.edge_inlier[e] = ((uint8_t*)(h + o[9]))[k];
            if (q.edge_chi2) q.edge_chi2[e] = ((double*)(h + o[17]))[k];
        }
        memcpy(q.stats, h + o[18], sizeof(int) * 4);
        if (q.stats[0] < 0) {
            suo_set_error("suo_optimize: %d free objects with free cameras exceeds the Schur limit of 16", q.n_obj);
            return SUO_ERR_ARG;
        }
    }
    return SUO_OK;
}

static int optimize_phasewise(suo_ba_problem* q);
static int optimize_phases_one_rank(suo_ba_problem* q);

// csrc/lm_frame2.hip takes one fixed camera, <= 16 objects, the edges that fit its LDS allotment, and -- its lanes keep their own edges'
// outlier flags in a 32-bit mask -- at most 32 edges per lane: 32 * G per object, G = 8 lanes (<= 8 objects) or 4 (9-16)
static bool frame2_takes(const suo_ba_problem& q) {
    if (q.n_cam != 1 || q.n_obj < 1 || q.n_obj > 16 || q.n_edge > lm_frame2_max_edges()) return false;
    int per_obj[16] = {0};
    for (int e = 0; e < q.n_edge; ++e) {
        const int o = q.edge_obj[e];
        if (o < 0 || o >= q.n_obj) return false;
        ++per_obj[o];
    }
    const int cap = 32 * (q.n_obj <= 8 ? 8 : 4);
    for (int o = 0; o < q.n_obj; ++o) if (per_obj[o] > cap) return false;
    return true;
}

int suo_optimize_batch(suo_ba_problem* probs, int n_prob) {
    if (n_prob <= 0) return SUO_OK;
    if (!probs) { suo_set_error("suo_optimize_batch: null argument"); return SUO_ERR_ARG; }
    // more than 16 free objects next to free cameras (T-LESS scenes): the reduced system outgrows the single-kernel paths;
    // those graphs run the phase kernels under the host schedule, one by one, the rest of the batch as usual
    {
        std::vector<int> small;
        bool any_big = false;
        for (int i = 0; i < n_prob; ++i) {
            int nfo = 0, nfc = 0;
            for (int o = 0; o < probs[i].n_obj; ++o) nfo += probs[i].obj_fixed[o] ? 0 : 1;
            for (int c = 0; c < probs[i].n_cam; ++c) nfc += probs[i].cam_fixed[c] ? 0 : 1;
            if (nfo > 16 && nfc > 0) any_big = true; else small.push_back(i);
        }
        if (any_big) {
            for (int i = 0, k = 0; i < n_prob; ++i) {
                if (k < (int)small.size() && small[k] == i) { ++k; int rc = suo_optimize_batch(&probs[i], 1); if (rc != SUO_OK) return rc; }
                else { int rc = optimize_phasewise(&probs[i]); if (rc != SUO_OK) return rc; }
            }
            return SUO_OK;
        }
    }
    // ONE large graph with free cameras and free objects (the global SLAM adjustment): the phase kernels of csrc/lm_dist.hip under the device-resident LM schedule,
    // driven from here (round 6).  Measured at 60 cameras x 8 objects: 106 us per LM trial against 129 for lm_grid_kernel's grid barriers; the Python-driven form of
    // this very schedule (suo_slam_amd/ba_dist.py, one rank) was already the faster route and ObjectSLAM.optimize could not reach it through one C call.
    // SUO_LM_PHASES (tuning builds, which also link csrc/lm_grid.hip): 0 = lm_grid_kernel as in rounds 4-5.
    {
        static const int phases = (int)SUO_TUNE("SUO_LM_PHASES", 1);
        static const int big_from_ph = (int)SUO_TUNE("SUO_LM_BIG_EDGES", 512);
        if (phases && n_prob == 1 && probs[0].n_edge >= big_from_ph) {
            int nfo = 0, nfc = 0;
            for (int o = 0; o < probs[0].n_obj; ++o) nfo += probs[0].obj_fixed[o] ? 0 : 1;
            for (int c = 0; c < probs[0].n_cam; ++c) nfc += probs[0].cam_fixed[c] ? 0 : 1;
            if (nfo > 0 && nfc > 0) return optimize_phases_one_rank(&probs[0]);
        }
    }
    std::lock_guard<std::mutex> lock(g_arena.mu);
    Staged st;
    int rc = stage_problems(probs, n_prob, g_arena, st);
    if (rc != SUO_OK) return rc;
    // frame-sized graphs: one 256-thread workgroup each (csrc/lm.hip); large graphs that the phase route above does not take (several in one call, or no free
    // object / no free camera): the 1024-thread single-workgroup build (csrc/lm_big.hip).  Tuning builds with SUO_LM_PHASES=0: ONE large graph spread over up
    // to 32 workgroups with grid barriers (csrc/lm_grid.hip), rounds 4-5's route.
    int max_edges = 0;
    for (int i = 0; i < n_prob; ++i) max_edges = std::max(max_edges, probs[i].n_edge);
    static const int big_from = (int)SUO_TUNE("SUO_LM_BIG_EDGES", 512);       // (640 edges: 9.5 vs 12.2 ms, 1000: 10.9 vs 16.0, 350: 8.6 vs 6.1)
#ifdef SUO_TUNING
    static const int grid_wgs = (int)SUO_TUNE("SUO_LM_GRID_WGS", 32);          // 0: never use the grid kernel
#endif
    // camera tracking (ObjectSLAM.optimize(curr_only=True)): one free camera, every object fixed -> one wave per problem
    static const int cam_kernel = (int)SUO_TUNE("SUO_LM_CAM", 1);                    // 0: general kernel (A/B)
    bool cam_only = cam_kernel != 0;
    for (int i = 0; i < n_prob && cam_only; ++i) {
        int nfc = 0, nfo = 0;
        for (int c = 0; c < probs[i].n_cam; ++c) nfc += probs[i].cam_fixed[c] ? 0 : 1;
        for (int o = 0; o < probs[i].n_obj; ++o) nfo += probs[i].obj_fixed[o] ? 0 : 1;
        cam_only = nfc == 1 && nfo == 0;
    }
    // single-view frames (evaluate.py --nviews 1): no free camera -> block-diagonal system, one wave per object (csrc/lm_frame.hip)
    static const int frame_kernel = (int)SUO_TUNE("SUO_LM_FRAME", 8);             // max objects per frame it takes; 0: off (A/B)
    bool frame_only = frame_kernel > 0 && !cam_only;
    int frame_max_obj = 0;
    for (int i = 0; i < n_prob && frame_only; ++i) {
        int nfc = 0;
        for (int c = 0; c < probs[i].n_cam; ++c) nfc += probs[i].cam_fixed[c] ? 0 : 1;
        // (one wave per object takes <= SUO_LM_FRAME objects: the 16-wave build spills; one wave per frame takes 16)
        const bool f2 = ((int)SUO_TUNE("SUO_LM_FRAME2", 1) != 0 && frame2_takes(probs[i]));
        frame_only = nfc == 0 && probs[i].n_obj >= 1 && probs[i].n_obj <= (f2 ? 16 : frame_kernel) && probs[i].n_obj <= 16;
        frame_max_obj = std::max(frame_max_obj, probs[i].n_obj);
    }
    if (cam_only) {
        // the camera alone in its graph (what ObjectSLAM.optimize(curr_only=True) builds): registers / LDS only (csrc/lm_cam2.hip)
        static const int cam2 = (int)suo::env_switch("SUO_LM_CAM2", 1);                  // 0: csrc/lm_cam.hip (A/B)
        bool alone = cam2 != 0;
        for (int i = 0; i < n_prob && alone; ++i) alone = probs[i].n_cam == 1 && probs[i].n_edge <= lm_cam2_max_edges();
        if (alone) rc = launch_lm_cam2(g_arena.dev + st.o_structs, n_prob, max_edges, g_arena.stream);
        else rc = launch_lm_cam(g_arena.dev + st.o_structs, n_prob, g_arena.stream);
    } else if (frame_only) {
        // one fixed camera (the single-view frame of evaluate.py): one WAVE per frame, the objects side by side (csrc/lm_frame2.hip)
        static const int frame2 = (int)SUO_TUNE("SUO_LM_FRAME2", 1);              // 0: one wave per object (A/B)
        bool one_cam = frame2 != 0;
        for (int i = 0; i < n_prob && one_cam; ++i) one_cam = frame2_takes(probs[i]);
        if (one_cam) rc = launch_lm_frame2(g_arena.dev + st.o_structs, n_prob, frame_max_obj, max_edges, g_arena.stream);
        else rc = launch_lm_frame(g_arena.dev + st.o_structs, n_prob, frame_max_obj, g_arena.stream);
#ifdef SUO_TUNING
    } else if (max_edges >= big_from && n_prob == 1 && grid_wgs > 0) {
        static void* grid_scratch = nullptr;
        if (!grid_scratch) SUO_HIP_CHECK(hipMalloc(&grid_scratch, lm_grid_scratch_bytes()));
        SUO_HIP_CHECK(hipMemsetAsync(grid_scratch, 0, 64, g_arena.stream));
        const int wgs = std::max(1, std::min(grid_wgs, (max_edges + 255) / 256));
        rc = launch_lm_grid(g_arena.dev + st.o_structs, grid_scratch, wgs, g_arena.stream);
#endif
    } else if (max_edges >= big_from) {
        rc = launch_lm_big(g_arena.dev + st.o_structs, n_prob, st.lds_need, g_arena.stream);
    } else {
        rc = launch_lm(g_arena.dev + st.o_structs, n_prob, st.lds_need, g_arena.stream);
    }
    if (rc != SUO_OK) return rc;
    return fetch_results(probs, n_prob, g_arena, st);
}

int suo_optimize(suo_ba_problem* problem) { return suo_optimize_batch(problem, 1); }

// ---- phase-wise bundle adjustment context (multi-GPU global BA; driver: suo_slam_amd/ba_dist.py) ------------
struct suo_ba_ctx {
    Arena arena;            // the device-resident problem (private: lives across calls)
    Staged st;
    int n_cam = 0, n_obj = 0, ns = 0;
    double* d_io = nullptr; double* h_io = nullptr; size_t io_doubles = 0;   // [out | in | workgroup partials (device only)]
    size_t io_cap = 0, big_cap = 0;                                           // capacities of the (possibly recycled) buffers, in doubles
    double* d_big = nullptr;     // reduced system + right-hand side in global memory when it has more than 96 rows (> 16 free objects)
    int device = 0;              // the device its buffers live on (hipGetDevice at creation)
    bool on_caller_stream = false;      // a *_dev entry has enqueued work on a stream this context does not own: destroy must not park the buffers under it
    hipStream_t on(void* stream) { on_caller_stream = true; return (hipStream_t)stream; }
    double* scratch() const { return d_io + io_doubles; }
    const void* dev_problem() const { return arena.dev + st.o_structs; }
};

// A context's buffers outlive it: a global adjustment of a SLAM run is create -> optimise -> destroy every few views, and creating / freeing its device arena, pinned
// staging, stream and exchange buffers cost 1.6-2 ms of a 10 ms adjustment (hipFree and hipHostFree synchronise the device).  suo_ba_ctx_destroy parks them here (at most
// four sets), suo_ba_ctx_create takes the first set OF ITS DEVICE back and grows what is too small.  Nothing read from them relies on their previous contents.
// The phase entries (suo_ba_*_dev) run on the CALLER's stream: a context that has used one is parked only after the whole device has drained (hipFree / hipHostFree
// used to imply that), so the next context never restages buffers a queued kernel still reads.
struct BaCtxBuffers {
    int device = -1;
    char* dev = nullptr; char* host = nullptr; size_t cap = 0; hipStream_t stream = nullptr;
    double* d_io = nullptr; double* h_io = nullptr; size_t io_cap = 0;      // io_cap: doubles of h_io; d_io holds io_cap + ba_scratch_doubles()
    double* d_big = nullptr; size_t big_cap = 0;
};
static std::mutex g_ba_pool_mu;
static std::vector<BaCtxBuffers> g_ba_pool;
static void ba_buffers_free(BaCtxBuffers& b) {
    if (b.d_io) (void)hipFree(b.d_io);
    if (b.d_big) (void)hipFree(b.d_big);
    if (b.h_io) (void)hipHostFree(b.h_io);
    if (b.dev) (void)hipFree(b.dev);
    if (b.host) (void)hipHostFree(b.host);
    if (b.stream) (void)hipStreamDestroy(b.stream);
    b = BaCtxBuffers();
}

int suo_ba_ctx_create(suo_ba_problem* p, suo_ba_ctx** out) {
    if (!p || !out) { suo_set_error("suo_ba_ctx_create: null argument"); return SUO_ERR_ARG; }
    suo_ba_ctx* c = new suo_ba_ctx();
    BaCtxBuffers b;
    if (hipGetDevice(&c->device) != hipSuccess) { delete c; suo_set_error("suo_ba_ctx_create: no current device"); return SUO_ERR_HIP; }
    {
        std::lock_guard<std::mutex> lock(g_ba_pool_mu);
        for (size_t i = g_ba_pool.size(); i-- > 0;)
            if (g_ba_pool[i].device == c->device) { b = g_ba_pool[i]; g_ba_pool.erase(g_ba_pool.begin() + i); break; }
    }
    c->arena.dev = b.dev; c->arena.host = b.host; c->arena.cap = b.cap; c->arena.stream = b.stream;      // (Arena::ensure keeps what is large enough)
    c->d_io = b.d_io; c->h_io = b.h_io; c->io_cap = b.io_cap; c->d_big = b.d_big; c->big_cap = b.big_cap;
    int rc = stage_problems(p, 1, c->arena, c->st);
    if (rc != SUO_OK) { suo_ba_ctx_destroy(c); return rc; }
    c->n_cam = p->n_cam; c->n_obj = p->n_obj;
    int nfo = 0;
    for (int o = 0; o < p->n_obj; ++o) nfo += p->obj_fixed[o] ? 0 : 1;
    c->ns = 6 * nfo;
    const size_t big_need = nfo > 16 ? (size_t)c->ns * c->ns + c->ns : 0;
    if (big_need > c->big_cap) {
        if (c->d_big) (void)hipFree(c->d_big);
        c->d_big = nullptr; c->big_cap = 0;
        if (hipMalloc((void**)&c->d_big, big_need * sizeof(double)) != hipSuccess) { suo_set_error("suo_ba_ctx_create: allocation failed"); suo_ba_ctx_destroy(c); return SUO_ERR_HIP; }
        c->big_cap = big_need;
    }
    c->io_doubles = 2 * ((size_t)c->ns * c->ns + c->ns + 27 * (size_t)p->n_obj + 16);
    if (c->io_doubles > c->io_cap) {
        if (c->d_io) (void)hipFree(c->d_io);
        if (c->h_io) (void)hipHostFree(c->h_io);
        c->d_io = nullptr; c->h_io = nullptr; c->io_cap = 0;
        if (hipMalloc((void**)&c->d_io, (c->io_doubles + ba_scratch_doubles()) * sizeof(double)) != hipSuccess ||
            hipHostMalloc((void**)&c->h_io, c->io_doubles * sizeof(double), hipHostMallocDefault) != hipSuccess) {
            suo_set_error("suo_ba_ctx_create: allocation failed"); suo_ba_ctx_destroy(c); return SUO_ERR_HIP;
        }
        c->io_cap = c->io_doubles;
    }
    rc = launch_ba_init(c->dev_problem(), c->arena.stream);
    if (rc != SUO_OK) { suo_ba_ctx_destroy(c); return rc; }
    SUO_HIP_CHECK(hipStreamSynchronize(c->arena.stream));
    *out = c;
    return SUO_OK;
}

void suo_ba_ctx_destroy(suo_ba_ctx* c) {
    if (!c) return;
    int cur = -1;
    const bool switched = hipGetDevice(&cur) == hipSuccess && cur != c->device && hipSetDevice(c->device) == hipSuccess;
    if (c->on_caller_stream) (void)hipDeviceSynchronize();          // work may still be queued on a stream that is not ours
    else if (c->arena.stream) (void)hipStreamSynchronize(c->arena.stream);
    BaCtxBuffers b;
    b.device = c->device;
    b.dev = c->arena.dev; b.host = c->arena.host; b.cap = c->arena.cap; b.stream = c->arena.stream;
    b.d_io = c->d_io; b.h_io = c->h_io; b.io_cap = c->io_cap; b.d_big = c->d_big; b.big_cap = c->big_cap;
    c->arena.dev = nullptr; c->arena.host = nullptr; c->arena.stream = nullptr;
    delete c;
    static const size_t keep = (int)SUO_TUNE("SUO_BA_CTX_POOL", 4);      // 0: free at once
    bool parked = false;
    {
        std::lock_guard<std::mutex> lock(g_ba_pool_mu);
        if (g_ba_pool.size() < keep) { g_ba_pool.push_back(b); parked = true; }
    }
    if (!parked) ba_buffers_free(b);
    if (switched) (void)hipSetDevice(cur);
}

int suo_ba_ctx_ns(const suo_ba_ctx* c) { return c ? c->ns : -1; }

static int ba_fetch(suo_ba_ctx* c, double* out, size_t n) {
    SUO_HIP_CHECK(hipMemcpyAsync(c->h_io, c->d_io, n * sizeof(double), hipMemcpyDeviceToHost, c->arena.stream));
    SUO_HIP_CHECK(hipStreamSynchronize(c->arena.stream));
    memcpy(out, c->h_io, n * sizeof(double));
    return SUO_OK;
}

int suo_ba_classify(suo_ba_ctx* c, int keep_all, double* num_good_local) {
    int rc = launch_ba_classify(c->dev_problem(), keep_all, c->d_io, c->scratch(), c->arena.stream);
    return rc != SUO_OK ? rc : ba_fetch(c, num_good_local, 1);
}

int suo_ba_linearize(suo_ba_ctx* c, int robust_on, double* out) {
    int rc = launch_ba_linearize(c->dev_problem(), robust_on, c->d_io, c->scratch(), 0, 1, c->arena.stream);
    return rc != SUO_OK ? rc : ba_fetch(c, out, 2 + 27 * (size_t)c->n_obj);
}

int suo_ba_schur(suo_ba_ctx* c, double lambda, double* out) {
    int rc = launch_ba_schur(c->dev_problem(), lambda, c->ns, c->d_io, c->scratch(), c->arena.stream);
    return rc != SUO_OK ? rc : ba_fetch(c, out, (size_t)c->ns * c->ns + c->ns + 1);
}

int suo_ba_solve_update(suo_ba_ctx* c, double lambda, int robust_on, const double* in, double* out) {
    const size_t n_in = 27 * (size_t)c->n_obj + (size_t)c->ns * c->ns + c->ns;
    double* h_in = c->h_io + c->io_doubles / 2;
    double* d_in = c->d_io + c->io_doubles / 2;
    memcpy(h_in, in, n_in * sizeof(double));
    SUO_HIP_CHECK(hipMemcpyAsync(d_in, h_in, n_in * sizeof(double), hipMemcpyHostToDevice, c->arena.stream));
    int rc = launch_ba_solve_update(c->dev_problem(), lambda, c->ns, robust_on, d_in, d_in + 27 * (size_t)c->n_obj, 0, c->d_io, c->scratch(),
                                    c->d_big, c->arena.stream);
    if (rc != SUO_OK) return rc;
    double dev_order[4];                       // device layout [chi2 | scale_cams | ok | scale_objs] -> documented host layout
    rc = ba_fetch(c, dev_order, 4);
    out[0] = dev_order[0]; out[1] = dev_order[1]; out[2] = dev_order[3]; out[3] = dev_order[2];
    return rc;
}

int suo_ba_restore(suo_ba_ctx* c) {
    int rc = launch_ba_restore(c->dev_problem(), c->arena.stream);
    if (rc != SUO_OK) return rc;
    SUO_HIP_CHECK(hipStreamSynchronize(c->arena.stream));
    return SUO_OK;
}

// ---- graphs whose reduced system does not fit the single-kernel paths (more than 16 free objects next to free cameras) ----
// The phase kernels above under g2o's LM schedule (optimization_algorithm_levenberg.cpp:58-150) and the robust rounds of
// ObjectSLAM.optimize (lib/object_slam.py:842-896), driven from the host: the one-rank form of suo_slam_amd/ba_dist.py.
static int optimize_phasewise(suo_ba_problem* q) {
    suo_ba_ctx* c = nullptr;
    int rc = suo_ba_ctx_create(q, &c);
    if (rc != SUO_OK) return rc;
    struct Guard { suo_ba_ctx* c; ~Guard() { suo_ba_ctx_destroy(c); } } guard{c};
    const int O = q->n_obj, ns = c->ns;
    std::vector<double> lin(2 + 27 * (size_t)O), sch((size_t)ns * ns + ns + 1), tot(27 * (size_t)O + (size_t)ns * ns + ns);
    double good = 0, red[4];
    int rounds = 0, lm_its = 0, lm_trials = 0, num_good = q->n_edge;
    if (q->init_with_outliers) { rc = suo_ba_classify(c, 1, &good); if (rc) return rc; }
    else { rc = suo_ba_classify(c, 0, &good); if (rc) return rc; num_good = (int)(good + 0.5); }
    bool robust_on = true;
    const int drop = std::max(1, q->n_rounds / 2);
    static const int diag21[6] = {0, 6, 11, 15, 18, 20};
    for (int rnd = 0; rnd < q->n_rounds; ++rnd) {
        if (q->n_edge < 4 || num_good < 4) break;
        ++rounds;
        double lam = -1, ni = 2;
        for (int it = 0; it < q->its[rnd]; ++it) {
            rc = suo_ba_linearize(c, robust_on, lin.data()); if (rc) return rc;
            double current_chi = lin[0];
            const double* HB = lin.data() + 1;
            if (it == 0) {                                          // computeLambdaInit: tau * max |diag H| over all free vertices
                double maxd = lin[1 + 27 * (size_t)O];
                for (int o = 0; o < O; ++o)
                    if (!q->obj_fixed[o]) for (int d = 0; d < 6; ++d) maxd = std::max(maxd, fabs(HB[27 * o + diag21[d]]));
                lam = 1e-5 * maxd; ni = 2;
            }
            double rho = 0; int qmax = 0; bool lam_finite = true;
            do {
                rc = suo_ba_schur(c, lam, sch.data()); if (rc) return rc;
                double temp_chi = 1.7976931348623157e308, scale = 0;
                if (sch[(size_t)ns * ns + ns] > 0.5) {
                    memcpy(tot.data(), HB, 27 * (size_t)O * sizeof(double));
                    memcpy(tot.data() + 27 * (size_t)O, sch.data(), ((size_t)ns * ns + ns) * sizeof(double));
                    rc = suo_ba_solve_update(c, lam, robust_on, tot.data(), red); if (rc) return rc;
                    if (red[3] > 0.5) { temp_chi = red[0]; scale = red[1] + red[2]; }
                }
                rho = (current_chi - temp_chi) / (scale + 1e-3);
                if (rho > 0 && std::isfinite(temp_chi)) {
                    const double alpha = std::min(1.0 - pow(2 * rho - 1, 3.0), 2.0 / 3.0);
                    lam *= std::max(1.0 / 3.0, alpha); ni = 2; current_chi = temp_chi;
                } else {
                    lam *= ni; ni *= 2;
                    rc = suo_ba_restore(c); if (rc) return rc;
                    if (!std::isfinite(lam)) { lam_finite = false; break; }
                }
                ++qmax; ++lm_trials;
            } while (rho < 0 && qmax < 10);
            ++lm_its;
            if (qmax == 10 || rho == 0 || !lam_finite) break;
        }
        rc = suo_ba_classify(c, 0, &good); if (rc) return rc;
        num_good = (int)(good + 0.5);
        if (rnd == drop) robust_on = false;
    }
    rc = suo_ba_ctx_download(c, q);
    q->stats[0] = rounds; q->stats[1] = lm_its; q->stats[2] = lm_trials; q->stats[3] = num_good;
    return rc;
}

// One rank, device-resident schedule, driven from C: what suo_slam_amd/ba_dist.py: optimize_distributed does at world = 1 (units enqueued blindly, g2o's accept / reject
// arithmetic in the control block, the host looks at 16 doubles once per <= 12 units) without Python between the launches.  Exchange buffers: one grow-only device
// block + 16 pinned doubles, process-wide under their own lock.
static int optimize_phases_one_rank(suo_ba_problem* q) {
    static std::mutex mu;
    static double* d_buf = nullptr; static size_t d_cap = 0; static double* h_pin = nullptr; static int buf_dev = -1;
    std::lock_guard<std::mutex> lock(mu);
    suo_ba_ctx* c = nullptr;
    int rc = suo_ba_ctx_create(q, &c);
    if (rc != SUO_OK) return rc;
    struct Guard { suo_ba_ctx* c; ~Guard() { suo_ba_ctx_destroy(c); } } guard{c};
    const int O = q->n_obj, ns = c->ns;
    const size_t n_lin = 2 + 27 * (size_t)O, n_sch = (size_t)ns * ns + ns + 1;
    const size_t need = 2 * n_lin + n_sch + 4 + 1 + 16;
    if (need > d_cap || buf_dev != c->device) {
        if (d_buf) (void)hipFree(d_buf);
        d_buf = nullptr; d_cap = 0;
        SUO_HIP_CHECK(hipMalloc((void**)&d_buf, need * sizeof(double)));
        d_cap = need; buf_dev = c->device;
    }
    if (!h_pin) SUO_HIP_CHECK(hipHostMalloc((void**)&h_pin, 16 * sizeof(double), hipHostMallocPortable));
    hipStream_t s = c->arena.stream;
    SUO_HIP_CHECK(hipMemsetAsync(d_buf, 0, need * sizeof(double), s));
    double* lin_loc = d_buf; double* lin = lin_loc + n_lin; double* sch = lin + n_lin; double* red = sch + n_sch; double* good = red + 4; double* ctl = good + 1;
    auto look = [&](const double* dev, int n) -> int {
        SUO_HIP_CHECK(hipMemcpyAsync(h_pin, dev, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, s));
        SUO_HIP_CHECK(hipStreamSynchronize(s));
        return SUO_OK;
    };
    auto classify = [&](int keep_all, int* n_good) -> int {
        int r = launch_ba_classify(c->dev_problem(), keep_all, good, c->scratch(), s);
        if (r != SUO_OK) return r;
        r = look(good, 1);
        *n_good = (int)(h_pin[0] + 0.5);
        return r;
    };
    int rounds = 0, lm_its = 0, lm_trials = 0, num_good = q->n_edge, tmp = 0;
    if (q->init_with_outliers) { rc = classify(1, &tmp); if (rc) return rc; }
    else { rc = classify(0, &num_good); if (rc) return rc; }
    int robust_on = 1;
    const int drop = std::max(1, q->n_rounds / 2);
    static const int batch = (int)SUO_TUNE("SUO_BA_UNITS_PER_LOOK", 12);
    for (int rnd = 0; rnd < q->n_rounds; ++rnd) {
        if (q->n_edge < 4 || num_good < 4) break;
        ++rounds;
        const int its = q->its[rnd];
        rc = launch_ba_ctl_begin(ctl, its, 1, s); if (rc) return rc;
        int budget = std::min(its, batch);
        bool done = its <= 0;
        while (!done) {
            for (int u = 0; u < budget; ++u) {                // one unit = 12 launches, the control steps riding in the tail kernels (suo_ba_lm_unit_one_rank_dev)
                rc = launch_ba_linearize(c->dev_problem(), robust_on, lin_loc, c->scratch(), 0, 1, s, ctl, lin, 1 + 27 * O + 1, ctl); if (rc) return rc;
                rc = launch_ba_schur(c->dev_problem(), 0.0, ns, sch, c->scratch(), s, ctl); if (rc) return rc;
                rc = launch_ba_solve_update(c->dev_problem(), 0.0, ns, robust_on, lin + 1, sch, 1, red, c->scratch(), c->d_big, s, ctl, ctl); if (rc) return rc;
            }
            rc = look(ctl, 16); if (rc) return rc;
            done = (int)h_pin[3] == 2;
            budget = std::min(std::max(1, its - (int)h_pin[4]) + 1, batch);
        }
        lm_its = (int)h_pin[7]; lm_trials = (int)h_pin[8];
        rc = classify(0, &num_good); if (rc) return rc;
        if (rnd == drop) robust_on = 0;
    }
    rc = suo_ba_ctx_download(c, q);
    q->stats[0] = rounds; q->stats[1] = lm_its; q->stats[2] = lm_trials; q->stats[3] = num_good;
    return rc;
}

// ---- the same phases on caller-owned DEVICE buffers, stream-ordered, no host synchronisation: the buffers are what RCCL
// all-reduces in place between the phases (suo_slam_amd/ba_dist.py) ------------------------------------------------
int suo_ba_classify_dev(suo_ba_ctx* c, int keep_all, double* num_good_dev, void* stream) {
    return launch_ba_classify(c->dev_problem(), keep_all, num_good_dev, c->scratch(), c->on(stream));
}
int suo_ba_linearize_dev(suo_ba_ctx* c, int robust_on, int rank, int world, double* lin_dev, void* stream) {
    if (rank < 0 || rank >= world) { suo_set_error("suo_ba_linearize_dev: rank %d of %d", rank, world); return SUO_ERR_ARG; }
    return launch_ba_linearize(c->dev_problem(), robust_on, lin_dev, c->scratch(), rank, world, c->on(stream));
}
int suo_ba_schur_dev(suo_ba_ctx* c, double lambda, double* sch_dev, void* stream) {
    return launch_ba_schur(c->dev_problem(), lambda, c->ns, sch_dev, c->scratch(), c->on(stream));
}
int suo_ba_solve_update_dev(suo_ba_ctx* c, double lambda, int robust_on, int world, const double* lin_dev, const double* sch_dev, double* red_dev,
                            void* stream) {
    return launch_ba_solve_update(c->dev_problem(), lambda, c->ns, robust_on, lin_dev + 1, sch_dev, world, red_dev, c->scratch(), c->d_big,
                                  c->on(stream));
}
int suo_ba_restore_dev(suo_ba_ctx* c, void* stream) { return launch_ba_restore(c->dev_problem(), c->on(stream)); }

// ---- the same phases under the device-resident LM schedule (csrc/lm_dist.hip: ctl) ------------------------------------------------------
int suo_ba_lm_begin_dev(suo_ba_ctx* c, double* ctl_dev, int its, int world, void* stream) {
    if (!c || !ctl_dev || world < 1) { suo_set_error("suo_ba_lm_begin_dev: bad arguments"); return SUO_ERR_ARG; }
    return launch_ba_ctl_begin(ctl_dev, its, world, c->on(stream));
}
int suo_ba_lm_linearize_dev(suo_ba_ctx* c, int robust_on, int rank, int world, const double* ctl_dev, double* lin_local_dev, double* lin_dev, void* stream) {
    if (!c || !ctl_dev || !lin_local_dev || !lin_dev || rank < 0 || rank >= world) { suo_set_error("suo_ba_lm_linearize_dev: bad arguments"); return SUO_ERR_ARG; }
    // the reduce works in place and runs every unit: it starts from this rank's own totals every time (a trial on a standing linearisation
    // re-reduces the same numbers instead of reducing the already reduced ones) -- the tail kernel copies them over, live unit or not
    return launch_ba_linearize(c->dev_problem(), robust_on, lin_local_dev, c->scratch(), rank, world, c->on(stream), ctl_dev, lin_dev, 1 + 27 * c->n_obj + world);
}
int suo_ba_lm_schur_dev(suo_ba_ctx* c, double* ctl_dev, const double* lin_dev, double* sch_dev, void* stream) {
    if (!c || !ctl_dev || !lin_dev || !sch_dev) { suo_set_error("suo_ba_lm_schur_dev: null argument"); return SUO_ERR_ARG; }
    int rc = launch_ba_ctl_lin(c->dev_problem(), ctl_dev, lin_dev, c->scratch(), c->on(stream));
    if (rc != SUO_OK) return rc;
    return launch_ba_schur(c->dev_problem(), 0.0, c->ns, sch_dev, c->scratch(), c->on(stream), ctl_dev);
}
int suo_ba_lm_solve_update_dev(suo_ba_ctx* c, int robust_on, int world, const double* ctl_dev, const double* lin_dev, const double* sch_dev, double* red_dev,
                               void* stream) {
    if (!c || !ctl_dev || !lin_dev || !sch_dev || !red_dev) { suo_set_error("suo_ba_lm_solve_update_dev: null argument"); return SUO_ERR_ARG; }
    return launch_ba_solve_update(c->dev_problem(), 0.0, c->ns, robust_on, lin_dev + 1, sch_dev, world, red_dev, c->scratch(), c->d_big, c->on(stream),
                                  ctl_dev);
}
// One unit on ONE rank (no exchange between its phases): the control steps run inside the tail kernels in front of them -- 12 launches instead of 14.  Same arithmetic
// in the same order as the four calls above with nothing in between: bit-identical (tests/test_gpu_geometry.py).
int suo_ba_lm_unit_one_rank_dev(suo_ba_ctx* c, int robust_on, double* ctl_dev, double* lin_local_dev, double* lin_dev, double* sch_dev, double* red_dev, void* stream) {
    if (!c || !ctl_dev || !lin_local_dev || !lin_dev || !sch_dev || !red_dev) { suo_set_error("suo_ba_lm_unit_one_rank_dev: null argument"); return SUO_ERR_ARG; }
    hipStream_t s = c->on(stream);
    int rc = launch_ba_linearize(c->dev_problem(), robust_on, lin_local_dev, c->scratch(), 0, 1, s, ctl_dev, lin_dev, 1 + 27 * c->n_obj + 1, ctl_dev);
    if (rc != SUO_OK) return rc;
    rc = launch_ba_schur(c->dev_problem(), 0.0, c->ns, sch_dev, c->scratch(), s, ctl_dev);
    if (rc != SUO_OK) return rc;
    return launch_ba_solve_update(c->dev_problem(), 0.0, c->ns, robust_on, lin_dev + 1, sch_dev, 1, red_dev, c->scratch(), c->d_big, s, ctl_dev, ctl_dev);
}
int suo_ba_lm_decide_dev(suo_ba_ctx* c, double* ctl_dev, const double* red_dev, void* stream) {
    if (!c || !ctl_dev || !red_dev) { suo_set_error("suo_ba_lm_decide_dev: null argument"); return SUO_ERR_ARG; }
    return launch_ba_ctl_decide(c->dev_problem(), ctl_dev, red_dev, c->on(stream));
}

// Test entry: the workgroup Cholesky solve of the reduced system (csrc/lm_device.h: wg_cholesky_solve) on a dense symmetric ns x ns matrix (host, row-major; ns a multiple
// of 6, at most 96) -- x solves A x = b; *ok_out = 0 when a pivot was not positive (x is then meaningless).
int suo_debug_cholesky_solve(const double* A, const double* b, int ns, double* x_out, int* ok_out) {
    if (!A || !b || !x_out || !ok_out || ns <= 0) { suo_set_error("suo_debug_cholesky_solve: bad argument"); return SUO_ERR_ARG; }
    double* d = nullptr;
    const size_t n = (size_t)ns * ns + 2 * (size_t)ns + 1;
    SUO_HIP_CHECK(hipMalloc((void**)&d, n * sizeof(double)));
    struct Free { double* d; ~Free() { (void)hipFree(d); } } guard{d};
    SUO_HIP_CHECK(hipMemcpy(d, A, (size_t)ns * ns * sizeof(double), hipMemcpyHostToDevice));
    SUO_HIP_CHECK(hipMemcpy(d + (size_t)ns * ns, b, (size_t)ns * sizeof(double), hipMemcpyHostToDevice));
    int rc = launch_debug_cholesky(d, d + (size_t)ns * ns, ns, d + (size_t)ns * ns + ns, (int*)(d + (size_t)ns * ns + 2 * (size_t)ns), nullptr);
    if (rc != SUO_OK) return rc;
    SUO_HIP_CHECK(hipDeviceSynchronize());
    SUO_HIP_CHECK(hipMemcpy(x_out, d + (size_t)ns * ns + ns, (size_t)ns * sizeof(double), hipMemcpyDeviceToHost));
    SUO_HIP_CHECK(hipMemcpy(ok_out, d + (size_t)ns * ns + 2 * (size_t)ns, sizeof(int), hipMemcpyDeviceToHost));
    return SUO_OK;
}

// Test entry: what the LM kernels linearise.  After suo_ba_linearize (edge_pass_partial of csrc/lm_device.h, shared by every LM
// kernel) the context holds, per edge in the CALLER's edge order: jac[29] = [Jc 2x6 | Jo 2x6 | w*info (xx,xy,yy) | -w*info*err (2)]
// and err[2].  Inactive edges (outliers, fixed-fixed) keep whatever was there before: call it on all-inlier graphs.
int suo_debug_ba_jacobians(suo_ba_ctx* c, int n_edge, double* jac_out, double* err_out) {
    if (!c || !jac_out || !err_out) { suo_set_error("suo_debug_ba_jacobians: null argument"); return SUO_ERR_ARG; }
    const Prep& P = c->st.prep[0];
    if ((int)P.order.size() != n_edge) { suo_set_error("suo_debug_ba_jacobians: context has %d edges", (int)P.order.size()); return SUO_ERR_ARG; }
    std::vector<double> jac((size_t)29 * n_edge), err((size_t)2 * n_edge);
    SUO_HIP_CHECK(hipStreamSynchronize(c->arena.stream));
    SUO_HIP_CHECK(hipMemcpy(jac.data(), c->arena.dev + P.o[36], jac.size() * sizeof(double), hipMemcpyDeviceToHost));
    SUO_HIP_CHECK(hipMemcpy(err.data(), c->arena.dev + P.o[23], err.size() * sizeof(double), hipMemcpyDeviceToHost));
    for (int k = 0; k < n_edge; ++k) {
        const int e = P.order[k];
        memcpy(jac_out + (size_t)29 * e, jac.data() + (size_t)29 * k, 29 * sizeof(double));
        err_out[2 * e] = err[2 * k]; err_out[2 * e + 1] = err[2 * k + 1];
    }
    return SUO_OK;
}

int suo_ba_ctx_download(suo_ba_ctx* c, suo_ba_problem* p) {
    int rc = launch_ba_finalize(c->dev_problem(), c->arena.stream);
    if (rc != SUO_OK) return rc;
    return fetch_results(p, 1, c->arena, c->st);
}

}  // extern "C"
