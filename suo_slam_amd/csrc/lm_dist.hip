// Phase kernels of the multi-GPU global bundle adjustment (third row; BASELINE config 5).
//
// The global pose graph of ObjectSLAM.optimize (/root/reference/lib/object_slam.py:703-903) is bipartite
// (camera <-> object edges only, :821-823) with cameras >> objects.  Cameras (views) are partitioned across
// GPUs; every GPU holds all object poses and the edges of its own cameras.  One LM trial is then
//     local:   linearise own edges, eliminate own cameras -> S_g = sum_c Hco^T (Hcc+lambda I)^-1 Hco, r_g
//     RCCL:    all-reduce(sum) of [S_g | r_g]  ((6 n_obj)^2 + 6 n_obj doubles: 74.5 KB at 16 objects)
//     local:   every rank solves the SAME reduced system (blockdiag(Hoo + lambda I) - S) x_o = b_o - r,
//              back-substitutes its own cameras, updates poses, evaluates its part of chi2
//     RCCL:    all-reduce(sum) of [chi2, step-scale] -> identical accept / reject and lambda on every rank
// which is exactly the system the single-GPU kernel (csrc/lm.hip) forms in LDS; g2o's lambda schedule
// (optimization_algorithm_levenberg.cpp:58-150) runs on the host between phases (suo_slam_amd/ba_dist.py).
// Each phase is one single-workgroup launch over the device-resident problem (same LmProblem SoA).
#include "lm_device.h"

namespace suo {

constexpr int DIAG21[6] = {0, 6, 11, 15, 18, 20};

// ---- phase 0: poses from the 3x4 matrices, reset levels ---------------------------------------------------
__global__ __launch_bounds__(LM_THREADS) void ba_init_kernel(const LmProblem* __restrict__ Pp) {
    const LmProblem& P = *Pp;
    const int tid = threadIdx.x;
    for (int c = tid; c < P.n_cam; c += LM_THREADS) pose_from_T(P.cam_T + 12 * c, P.cam[c]);
    for (int o = tid; o < P.n_obj; o += LM_THREADS) pose_from_T(P.obj_T + 12 * o, P.obj[o]);
    for (int e = tid; e < P.n_edge; e += LM_THREADS) P.level[e] = 0;
    if (tid == 0) {
        int ns = 0;
        for (int o = 0; o < P.n_obj; ++o) P.obj_slot[o] = P.obj_fixed[o] ? -1 : ns++;
    }
}

// ---- chi2 (re-)classification of the local edges (object_slam.py:855-866, 877-893); out[0] = local num_good ----
__global__ __launch_bounds__(LM_THREADS) void ba_classify_kernel(const LmProblem* __restrict__ Pp, int keep_all, double* __restrict__ out) {
    const LmProblem& P = *Pp;
    __shared__ double red[LM_THREADS / 64];
    const int tid = threadIdx.x;
    double good = 0;
    for (int e = tid; e < P.n_edge; e += LM_THREADS) {
        double er[2];
        edge_error(P, e, er, nullptr, nullptr);
        const double c2 = edge_chi2(P, e, er);
        P.edge_chi2[e] = c2;
        if (keep_all) { good += 1; continue; }            // opt_init_with_outliers and curr_only: nothing is cast out
        if (c2 > P.chi2_thr) { P.level[e] = 1; P.edge_inlier[e] = 0; }
        else { P.level[e] = 0; P.edge_inlier[e] = 1; good += 1; }
    }
    good = block_sum(good, red);
    if (tid == 0) out[0] = good;
}

// ---- linearise: out = [chi2_local | (Hoo 21 + bo 6) per object | max |diag Hcc| of local free cameras] -----
__global__ __launch_bounds__(LM_THREADS) void ba_linearize_kernel(const LmProblem* __restrict__ Pp, int robust_on, double* __restrict__ out) {
    const LmProblem& P = *Pp;
    __shared__ double red[LM_THREADS / 64];
    const int tid = threadIdx.x;
    const double chi = active_errors_and_chi2(P, robust_on != 0, true, red);
    accumulate_pairs(P);
    __syncthreads();
    for (int idx = tid; idx < P.n_cam * 27; idx += LM_THREADS) {
        const int c = idx / 27, k = idx - c * 27;
        if (P.cam_fixed[c]) continue;
        double s = 0;
        for (int j = P.cam_pair_ptr[c]; j < P.cam_pair_ptr[c + 1]; ++j)
            s += P.pair_part[90 * (size_t)P.cam_pair_idx[j] + (k < 21 ? k : 78 + (k - 21))];
        if (k < 21) P.Hcc[36 * c + k] = s; else P.bc[6 * c + (k - 21)] = s;
    }
    for (int idx = tid; idx < P.n_obj * 27; idx += LM_THREADS) {
        const int o = idx / 27, k = idx - o * 27;
        double s = 0;
        if (!P.obj_fixed[o])
            for (int j = P.obj_pair_ptr[o]; j < P.obj_pair_ptr[o + 1]; ++j)
                s += P.pair_part[90 * (size_t)P.obj_pair_idx[j] + (k < 21 ? 21 + k : 84 + (k - 21))];
        out[1 + idx] = s;
    }
    __syncthreads();
    double md = 0;
    for (int idx = tid; idx < P.n_cam * 6; idx += LM_THREADS) {
        const int c = idx / 6;
        if (!P.cam_fixed[c]) md = fmax(md, fabs(P.Hcc[36 * c + DIAG21[idx - c * 6]]));
    }
    md = block_max(md, red);
    if (tid == 0) { out[0] = chi; out[1 + 27 * P.n_obj] = md; }
}

// ---- local Schur complement for this lambda: out = [S_g (ns x ns) | r_g (ns) | ok] ; also push() ------------
__global__ __launch_bounds__(LM_THREADS) void ba_schur_kernel(const LmProblem* __restrict__ Pp, double lambda, int ns, double* __restrict__ out) {
    const LmProblem& P = *Pp;
    __shared__ int sh_ok;
    const int tid = threadIdx.x;
    for (int c = tid; c < P.n_cam; c += LM_THREADS) P.cam_bak[c] = P.cam[c];
    for (int o = tid; o < P.n_obj; o += LM_THREADS) P.obj_bak[o] = P.obj[o];
    if (tid == 0) sh_ok = 1;
    __syncthreads();
    for (int c = tid; c < P.n_cam; c += LM_THREADS) {
        if (P.cam_fixed[c]) continue;
        double A[36], Ai[36];
        unpack_sym21(P.Hcc + 36 * c, A);
        for (int d = 0; d < 6; ++d) A[d * 7] += lambda;
        if (!spd_inverse6(A, Ai)) { sh_ok = 0; for (int i = 0; i < 36; ++i) Ai[i] = 0; }
        for (int i = 0; i < 36; ++i) P.Hcc_inv[36 * c + i] = Ai[i];
        for (int r = 0; r < 6; ++r) {
            double s = 0;
            for (int k = 0; k < 6; ++k) s += Ai[r * 6 + k] * P.bc[6 * c + k];
            P.yc[6 * c + r] = s;
        }
    }
    __syncthreads();
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
    __syncthreads();
    for (int idx = tid; idx < ns * ns; idx += LM_THREADS) {
        const int row = idx / ns, col = idx - row * ns;
        const int s1 = row / 6, i = row - s1 * 6, s2 = col / 6, j = col - s2 * 6;
        int o1 = -1, o2 = -1;
        for (int o = 0; o < P.n_obj; ++o) { if (P.obj_slot[o] == s1) o1 = o; if (P.obj_slot[o] == s2) o2 = o; }
        double acc = 0;
        for (int a = P.obj_pair_ptr[o1]; a < P.obj_pair_ptr[o1 + 1]; ++a) {
            const int p1 = P.obj_pair_idx[a], c = P.pair_cam[p1];
            if (P.cam_fixed[c]) continue;
            int p2 = -1;
            for (int b = P.cam_pair_ptr[c]; b < P.cam_pair_ptr[c + 1]; ++b)
                if (P.pair_obj[P.cam_pair_idx[b]] == o2) { p2 = P.cam_pair_idx[b]; break; }
            if (p2 < 0) continue;
            const double* H1 = P.pair_part + 90 * (size_t)p1 + 42;
            const double* Y2 = P.Y + 36 * (size_t)p2;
            for (int k = 0; k < 6; ++k) acc += H1[k * 6 + i] * Y2[k * 6 + j];
        }
        out[idx] = acc;
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
        out[ns * ns + row] = acc;
    }
    __syncthreads();
    if (tid == 0) out[ns * ns + ns] = (double)sh_ok;
}

// ---- solve the reduced system (identical on every rank), back-substitute own cameras, update, re-evaluate ----
// in  = [Hoo_total(21)+bo_total(6) per object | S_total (ns x ns) | r_total (ns)]
// out = [chi2_local after the step | sum x_c (lambda x_c + b_c) over own cameras | same over objects | ok]
__global__ __launch_bounds__(LM_THREADS) void ba_solve_update_kernel(const LmProblem* __restrict__ Pp, double lambda, int ns, int robust_on,
                                                                      const double* __restrict__ in, double* __restrict__ out) {
    const LmProblem& P = *Pp;
    __shared__ double S[LM_NS * LM_NS];
    __shared__ double rhs[LM_NS], colbuf[LM_NS];
    __shared__ double red[LM_THREADS / 64];
    __shared__ int sh_ok;
    const int tid = threadIdx.x;
    const double* HB = in;
    const double* St = in + 27 * P.n_obj;
    const double* rt = St + ns * ns;
    if (tid == 0) sh_ok = 1;
    for (int idx = tid; idx < ns * ns; idx += LM_THREADS) S[idx] = -St[idx];
    __syncthreads();
    for (int idx = tid; idx < P.n_obj * 36; idx += LM_THREADS) {
        const int o = idx / 36, rc = idx - o * 36, r = rc / 6, cc = rc - r * 6;
        const int so = P.obj_slot[o];
        if (so < 0) continue;
        const int rr = r < cc ? r : cc, c2 = r < cc ? cc : r;
        const int packed = rr * 6 - rr * (rr - 1) / 2 + (c2 - rr);
        S[(6 * so + r) * ns + 6 * so + cc] += HB[27 * o + packed] + (r == cc ? lambda : 0.0);
    }
    for (int idx = tid; idx < P.n_obj * 6; idx += LM_THREADS) {
        const int o = idx / 6, r = idx - o * 6;
        if (P.obj_slot[o] >= 0) rhs[6 * P.obj_slot[o] + r] = HB[27 * o + 21 + r] - rt[6 * P.obj_slot[o] + r];
    }
    __syncthreads();
    for (int j = 0; j < ns; ++j) {                  // workgroup Cholesky, as in csrc/lm.hip
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
    for (int j = 0; j < ns; ++j) {
        if (tid == 0) rhs[j] = rhs[j] / S[j * ns + j];
        __syncthreads();
        const double yj = rhs[j];
        for (int i = j + 1 + tid; i < ns; i += LM_THREADS) rhs[i] -= S[i * ns + j] * yj;
        __syncthreads();
    }
    for (int j = ns - 1; j >= 0; --j) {
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
    const bool ok = sh_ok != 0;
    if (ok) {
        for (int c = tid; c < P.n_cam; c += LM_THREADS) if (!P.cam_fixed[c]) pose_oplus(P.cam[c], P.xc + 6 * c);
        for (int o = tid; o < P.n_obj; o += LM_THREADS) if (!P.obj_fixed[o]) pose_oplus(P.obj[o], P.xo + 6 * o);
    }
    __syncthreads();
    const double chi = active_errors_and_chi2(P, robust_on != 0, false, red);
    double sc_c = 0, sc_o = 0;
    if (ok) {
        for (int idx = tid; idx < P.n_cam * 6; idx += LM_THREADS)
            if (!P.cam_fixed[idx / 6]) sc_c += P.xc[idx] * (lambda * P.xc[idx] + P.bc[idx]);
        for (int idx = tid; idx < P.n_obj * 6; idx += LM_THREADS)
            if (!P.obj_fixed[idx / 6]) sc_o += P.xo[idx] * (lambda * P.xo[idx] + HB[27 * (idx / 6) + 21 + (idx % 6)]);
    }
    sc_c = block_sum(sc_c, red);
    sc_o = block_sum(sc_o, red);
    if (tid == 0) { out[0] = chi; out[1] = sc_c; out[2] = sc_o; out[3] = ok ? 1.0 : 0.0; }
}

// ---- pop(): restore the poses saved by ba_schur_kernel ---------------------------------------------------
__global__ __launch_bounds__(LM_THREADS) void ba_restore_kernel(const LmProblem* __restrict__ Pp) {
    const LmProblem& P = *Pp;
    for (int c = threadIdx.x; c < P.n_cam; c += LM_THREADS) P.cam[c] = P.cam_bak[c];
    for (int o = threadIdx.x; o < P.n_obj; o += LM_THREADS) P.obj[o] = P.obj_bak[o];
}

__global__ __launch_bounds__(LM_THREADS) void ba_finalize_kernel(const LmProblem* __restrict__ Pp) {
    const LmProblem& P = *Pp;
    for (int c = threadIdx.x; c < P.n_cam; c += LM_THREADS) pose_to_T(P.cam[c], P.cam_T + 12 * c);
    for (int o = threadIdx.x; o < P.n_obj; o += LM_THREADS) pose_to_T(P.obj[o], P.obj_T + 12 * o);
}

#define BA_LAUNCH(k, ...)                                                                 \
    hipLaunchKernelGGL(k, dim3(1), dim3(LM_THREADS), 0, s, (const LmProblem*)P, ##__VA_ARGS__); \
    SUO_HIP_CHECK(hipGetLastError());                                                     \
    return SUO_OK;

int launch_ba_init(const void* P, hipStream_t s) { BA_LAUNCH(ba_init_kernel) }
int launch_ba_classify(const void* P, int keep_all, double* out, hipStream_t s) { BA_LAUNCH(ba_classify_kernel, keep_all, out) }
int launch_ba_linearize(const void* P, int robust_on, double* out, hipStream_t s) { BA_LAUNCH(ba_linearize_kernel, robust_on, out) }
int launch_ba_schur(const void* P, double lambda, int ns, double* out, hipStream_t s) { BA_LAUNCH(ba_schur_kernel, lambda, ns, out) }
int launch_ba_solve_update(const void* P, double lambda, int ns, int robust_on, const double* in, double* out, hipStream_t s) {
    BA_LAUNCH(ba_solve_update_kernel, lambda, ns, robust_on, in, out)
}
int launch_ba_restore(const void* P, hipStream_t s) { BA_LAUNCH(ba_restore_kernel) }
int launch_ba_finalize(const void* P, hipStream_t s) { BA_LAUNCH(ba_finalize_kernel) }

}  // namespace suo
