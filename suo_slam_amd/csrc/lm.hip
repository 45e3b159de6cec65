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
//   * edges are grouped by (camera, object) pair; threads linearise edges in parallel (error, analytic
//     Jacobians, Huber weight), then one thread per (pair, entry) sums the 6x6 blocks H_cc, H_oo, H_co and
//     the gradients over the pair's edges in edge order -- deterministic, no cross-lane reduction;
//   * the working set (poses, blocks, edges, Jacobians) is relocated to LDS when it fits (a single-view
//     frame always does), so the ~80 LM trials of a frame never leave the CU;
//   * cameras are eliminated by Schur complement (block-diagonal H_cc, 6x6 inverses, one thread each),
//     the reduced object system lives in LDS and is factorised by a workgroup Cholesky; with no free
//     camera (single-view mode) the system is block diagonal and each object is solved by one thread;
//   * g2o's lambda schedule (tau = 1e-5, rho-gain update, nu doubling, <= 10 trials) runs on-device.
// Many problems (frames) run concurrently as independent workgroups.


#include "lm_device.h"

namespace suo {


constexpr int LM_LDS_BYTES = 150 * 1024;     // dynamic LDS per workgroup (160 KiB per CU on gfx950)
constexpr int LM_STAGE_EDGES = 384;          // LDS stage for the Jacobians of one batch of pairs (87 KiB), large graphs only

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

// optional phase timers (-DSUO_LM_PROFILE): cycles between consecutive LMPROF(i) marks are charged to section i
#ifdef SUO_LM_PROFILE
#define LMPROF(i) do { __syncthreads(); if (tid == 0) { const long long _t = clock64(); lmprof_acc[lmprof_last] += _t - lmprof_t; lmprof_t = _t; lmprof_last = (i); } } while (0)
#else
#define LMPROF(i) do { } while (0)
#endif

__global__ __launch_bounds__(LM_THREADS) void lm_kernel(const LmProblem* __restrict__ problems, int lds_bytes) {
#ifdef SUO_LM_PROFILE
    long long lmprof_acc[11] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, lmprof_t = clock64();
    int lmprof_last = 10;
#endif
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
    const int sp = ns | 1;                  // odd pitch of S in LDS (lm_device.h: wg_cholesky_solve)
    if (schur) {
        off = (((size_t)ns * sp + 8) * sizeof(double) + 15) & ~(size_t)15;
        rhs = (double*)(lm_lds + off); off += ((size_t)ns * sizeof(double) + 15) & ~(size_t)15;
    }
    // Large graphs (the global SLAM adjustment): the [n_edge][29] Jacobians cannot live in LDS.  Instead of writing them
    // to HBM and re-reading each pair's rows 90 times from L2 (47 % of a 7500-edge adjustment), the edges are linearised
    // in batches of whole pairs straight into this LDS stage and the batch's pair blocks are summed from it.
    double* jac_stage = nullptr;
    if (P.n_edge > LM_STAGE_EDGES && off + (size_t)LM_STAGE_EDGES * 29 * sizeof(double) <= (size_t)lds_bytes) {
        jac_stage = (double*)(lm_lds + off);
        off += (size_t)LM_STAGE_EDGES * 29 * sizeof(double);
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
            LMPROF(0);
            double currentChi;
            if (!jac_stage) {
                currentChi = active_errors_and_chi2(P, robust_on, true, red);
                LMPROF(1);
                accumulate_pairs(P);
                __syncthreads();
            } else {
                double part = 0;
                for (int p0 = 0; p0 < P.n_pair;) {
                    if (tid == 0) {                             // batch = as many whole pairs as the stage holds
                        const int e0 = P.pair_start[p0];
                        int p1 = p0 + 1;
                        while (p1 < P.n_pair && P.pair_start[p1 + 1] - e0 <= LM_STAGE_EDGES) ++p1;
                        sh_flag = p1;
                    }
                    __syncthreads();
                    const int p1 = sh_flag, e0 = P.pair_start[p0], e1 = P.pair_start[p1];
                    LmProblem Q = P;
                    Q.jac = jac_stage - 29 * (size_t)e0;        // edge e of the batch -> stage row e - e0
                    part += edge_pass_partial(Q, e0, e1, robust_on, true);
                    __syncthreads();
                    accumulate_pairs_range(Q, p0, p1);
                    __syncthreads();
                    p0 = p1;
                }
                currentChi = block_sum(part, red);
            }
            LMPROF(2);
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
            LMPROF(3);
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
                LMPROF(4);
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
                    LMPROF(5);
                    // S = blockdiag(Hoo + lambda I);  rhs = b_o
                    for (int idx = tid; idx < ns * sp; idx += LM_THREADS) S[idx] = 0;
                    __syncthreads();
                    for (int idx = tid; idx < P.n_obj * 36; idx += LM_THREADS) {
                        const int o = idx / 36, rc = idx - o * 36, r = rc / 6, cc = rc - r * 6;
                        const int so = P.obj_slot[o];
                        if (so < 0) continue;
                        const int rr = r < cc ? r : cc, c2 = r < cc ? cc : r;
                        const int packed = rr * 6 - rr * (rr - 1) / 2 + (c2 - rr);
                        S[(6 * so + r) * sp + 6 * so + cc] = P.Hoo[36 * o + packed] + (r == cc ? lambda : 0.0);
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
                            const int p2 = P.cam_obj_pair[(size_t)c * P.n_obj + o2];     // the same camera's pair with object o2
                            if (P.cam_fixed[c] || p2 < 0) continue;
                            const double* H1 = P.pair_part + 90 * (size_t)p1 + 42;   // Hco(c,o1) [6x6], row = cam dof
                            const double* Y2 = P.Y + 36 * (size_t)p2;
                            for (int k = 0; k < 6; ++k) acc += H1[k * 6 + i] * Y2[k * 6 + j];
                        }
                        S[row * sp + col] -= acc;
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
                    LMPROF(6);
                    // Cholesky S = L L^T (lower, in place), then forward / backward substitution: blocked by 6 over the whole
                    // workgroup (lm_device.h: wg_cholesky_solve; odd LDS pitch sp against the 64-way bank conflict)
                    wg_cholesky_solve(S, sp, rhs, ns, tid, LM_THREADS, &sh_ok);
                    __syncthreads();
                    for (int idx = tid; idx < P.n_obj * 6; idx += LM_THREADS) {
                        const int o = idx / 6;
                        P.xo[idx] = P.obj_slot[o] >= 0 ? rhs[6 * P.obj_slot[o] + (idx - o * 6)] : 0.0;
                    }
                    __syncthreads();
                    LMPROF(7);
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
                LMPROF(8);
                const bool ok2 = sh_ok != 0;
                // update(x)
                if (ok2) {
                    for (int c = tid; c < P.n_cam; c += LM_THREADS) if (!P.cam_fixed[c]) pose_oplus(P.cam[c], P.xc + 6 * c);
                    for (int o = tid; o < P.n_obj; o += LM_THREADS) if (!P.obj_fixed[o]) pose_oplus(P.obj[o], P.xo + 6 * o);
                }
                __syncthreads();
                LMPROF(9);
                double tempChi = active_errors_and_chi2(P, robust_on, false, red);
                LMPROF(10);
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
#ifdef SUO_LM_PROFILE
    if (tid == 0 && P.n_edge >= 1000) {
        printf("lm profile (edges %d, cams %d): its %d trials %d; Mcycles per section:", P.n_edge, P.n_cam, lm_its, lm_trials);
        for (int i = 0; i < 11; ++i) printf(" [%d]%.2f", i, (double)lmprof_acc[i] * 1e-6);
        printf("\n");
    }
#endif
}

#ifdef SUO_LM_BIG
// the 1024-thread build of this file (csrc/lm_big.hip): only the kernel and its launcher
int launch_lm_big(const void* problems_dev, int n_problems, int lds_bytes, hipStream_t s) {
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
#else
// Dynamic LDS a problem of this size wants (everything resident), capped at LM_LDS_BYTES.  Small problems ask
// for little, so their workgroup can share a CU with the CNN's workgroups instead of waiting for an empty one.
int lm_lds_bytes(int C, int O, int E, int NP, int n_free_obj_schur) {
    size_t b = 0;
    const size_t ns = 6 * (size_t)n_free_obj_schur;
    b += 8 * (ns * (ns + 1) + 8 + ns) + 48;
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
#endif  // SUO_LM_BIG

}  // namespace suo
