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
// which is exactly the system the single-GPU kernels (csrc/lm.hip, csrc/lm_grid.hip) form; g2o's lambda schedule
// (optimization_algorithm_levenberg.cpp:58-150) runs on the host between phases (suo_slam_amd/ba_dist.py).
//
// A rank's share of a large graph is thousands of edges, so every data-parallel step of a phase is its own grid-strided
// launch over BA_WGS workgroups (the kernel boundary is the barrier between steps); sums are taken per workgroup in a
// fixed wave order and combined in workgroup order by a one-workgroup tail kernel (deterministic); only the reduced
// (<= 96 x 96) system is factorised by a single workgroup.
#include "lm_device.h"

namespace suo {

constexpr int DIAG21[6] = {0, 6, 11, 15, 18, 20};
constexpr int BA_WGS = 64;                     // workgroups of the grid-strided steps
// scratch (device, per context): [0, BA_WGS) workgroup partials | [BA_WGS] failure counter (as double bits of an int)
constexpr int BA_SCRATCH_DOUBLES = BA_WGS + 8;

#define GT (blockIdx.x * LM_THREADS + threadIdx.x)
#define GS (gridDim.x * LM_THREADS)

// ---- device-resident LM schedule (round 4) ------------------------------------------------------------------------------------------
// g2o's accept / reject logic (optimization_algorithm_levenberg.cpp:88-148) is branch-free arithmetic on a handful of all-reduced scalars.
// Instead of reading them back per trial, the host enqueues UNITS -- [linearise] -> reduce -> [lambda init] -> Schur -> reduce -> solve +
// update + chi2 -> reduce -> decide [+ restore] -- and a control block on the device says which of them are live:
//   ctl[0] lambda   [1] ni   [2] current chi2   [3] state (0: the next unit linearises; 1: the next unit is a trial on the standing
//   linearisation; 2: the round is over)   [4] iterations done in this round   [5] its iteration budget   [6] qmax   [7] LM iterations (all
//   rounds)   [8] LM trials (all rounds)   [9] rho   [10] restore flag of the unit   [11] ranks
// Every phase kernel takes (ctl, want): with ctl it returns at once unless ctl[3] == want and reads lambda from ctl[0].  A unit enqueued
// after the round ended is a handful of empty launches; the host looks at ctl once per batch of units, not once per trial.
constexpr int CTL_LAMBDA = 0, CTL_NI = 1, CTL_CHI = 2, CTL_STATE = 3, CTL_IT = 4, CTL_ITS = 5, CTL_QMAX = 6, CTL_LM_ITS = 7, CTL_TRIALS = 8, CTL_RHO = 9,
              CTL_RESTORE = 10, CTL_WORLD = 11;
constexpr int ST_LINEARIZE = 0, ST_TRIAL = 1, ST_DONE = 2;
#define BA_GUARD(ctl, want) do { if ((ctl) && (int)(ctl)[CTL_STATE] != (want)) return; } while (0)
__device__ void ba_ctl_lin(const LmProblem& P, double* __restrict__ ctl, const double* __restrict__ lin, int* __restrict__ bad);
__device__ void ba_decide(double* __restrict__ ctl, const double* __restrict__ red);
// ONE rank, no collective between the phases of a unit: the one-thread control steps fold into the one-workgroup tail kernels in front of them (fold_ctl of
// ba_linearize_tail_kernel / ba_update_tail_kernel; 14 -> 12 launches per unit).  With an all-reduce between tail and control step they stay launches of their own.

// ---- phase 0: poses from the 3x4 matrices, reset levels ---------------------------------------------------
__global__ __launch_bounds__(LM_THREADS) void ba_init_kernel(const LmProblem* __restrict__ Pp) {
    const LmProblem& P = *Pp;
    for (int c = GT; c < P.n_cam; c += GS) pose_from_T(P.cam_T + 12 * c, P.cam[c]);
    for (int o = GT; o < P.n_obj; o += GS) pose_from_T(P.obj_T + 12 * o, P.obj[o]);
    for (int e = GT; e < P.n_edge; e += GS) P.level[e] = 0;
    if (GT == 0) {
        int ns = 0;
        for (int o = 0; o < P.n_obj; ++o) P.obj_slot[o] = P.obj_fixed[o] ? -1 : ns++;
        for (int k = 0; k < 4; ++k) P.stats[k] = 0;      // the host schedule keeps the counters (ba_dist.py); download checks stats[0] >= 0
    }
}

// out[idx] = partial[0] + partial[1] + ... in workgroup order
__global__ void ba_sum_kernel(const double* __restrict__ partial, int n, double* __restrict__ out, int idx) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        double s = 0;
        for (int i = 0; i < n; ++i) s += partial[i];
        out[idx] = s;
    }
}

// ---- chi2 (re-)classification of the local edges (object_slam.py:855-866, 877-893) ---------------------------
__global__ __launch_bounds__(LM_THREADS) void ba_classify_kernel(const LmProblem* __restrict__ Pp, int keep_all, double* __restrict__ partial) {
    const LmProblem& P = *Pp;
    __shared__ double red[LM_THREADS / 64];
    double good = 0;
    for (int e = GT; e < P.n_edge; e += GS) {
        double er[2];
        edge_error(P, e, er, nullptr, nullptr);
        const double c2 = edge_chi2(P, e, er);
        P.edge_chi2[e] = c2;
        if (keep_all) { good += 1; continue; }            // opt_init_with_outliers and curr_only: nothing is cast out
        if (c2 > P.chi2_thr) { P.level[e] = 1; P.edge_inlier[e] = 0; }
        else { P.level[e] = 0; P.edge_inlier[e] = 1; good += 1; }
    }
    good = block_sum(good, red);
    if (threadIdx.x == 0) partial[blockIdx.x] = good;
}

// ---- linearise: edge pass | pair blocks | diagonal blocks | tail --------------------------------------------
__global__ __launch_bounds__(LM_THREADS) void ba_edge_pass_kernel(const LmProblem* __restrict__ Pp, int robust_on, int with_jac,
                                                                   double* __restrict__ partial, const double* __restrict__ ctl, int want) {
    BA_GUARD(ctl, want);
    const LmProblem& P = *Pp;
    __shared__ double red[LM_THREADS / 64];
    const double c = block_sum(edge_pass_partial(P, 0, P.n_edge, robust_on != 0, with_jac != 0, GT, GS), red);
    if (threadIdx.x == 0) partial[blockIdx.x] = c;
}
__global__ __launch_bounds__(LM_THREADS) void ba_accumulate_kernel(const LmProblem* __restrict__ Pp, const double* __restrict__ ctl, int want) {
    BA_GUARD(ctl, want);
    accumulate_pairs_range(*Pp, 0, Pp->n_pair, GT, GS);
}
// own cameras: Hcc / bc; objects: this rank's share of (Hoo 21 + bo 6) -> out[1 + 27 o + k]
// Eight lanes per entry, each walking every eighth pair of the vertex's list (two dependent index loads per pair are pure latency: one lane per entry took 17.5 us for
// 32 cameras x 16 objects), the eight partial sums combined by an xor butterfly -- the same tree on every lane, the same bits on every run.
__global__ __launch_bounds__(LM_THREADS) void ba_gather_kernel(const LmProblem* __restrict__ Pp, double* __restrict__ out, const double* __restrict__ ctl, int want) {
    BA_GUARD(ctl, want);
    const LmProblem& P = *Pp;
    constexpr int SL = 8;
    auto sum8 = [](double v) {
#pragma unroll
        for (int o = 1; o < SL; o <<= 1) v += __shfl_xor(v, o, 64);
        return v;
    };
    const int nc = P.n_cam * 27, no = P.n_obj * 27;
    // (whole groups of eight lanes run the same trip count: the shuffles are executed by all of them)
    for (int t = GT; t < (nc + no) * SL; t += GS) {
        const int item = t / SL, sl = t - item * SL;
        if (item < nc) {
            const int c = item / 27, k = item - c * 27;
            double s = 0;
            if (!P.cam_fixed[c])
                for (int j = P.cam_pair_ptr[c] + sl; j < P.cam_pair_ptr[c + 1]; j += SL)
                    s += P.pair_part[90 * (size_t)P.cam_pair_idx[j] + (k < 21 ? k : 78 + (k - 21))];
            s = sum8(s);
            if (sl == 0 && !P.cam_fixed[c]) { if (k < 21) P.Hcc[36 * c + k] = s; else P.bc[6 * c + (k - 21)] = s; }
        } else {
            const int idx = item - nc, o = idx / 27, k = idx - o * 27;
            double s = 0;
            if (!P.obj_fixed[o])
                for (int j = P.obj_pair_ptr[o] + sl; j < P.obj_pair_ptr[o + 1]; j += SL)
                    s += P.pair_part[90 * (size_t)P.obj_pair_idx[j] + (k < 21 ? 21 + k : 84 + (k - 21))];
            s = sum8(s);
            if (sl == 0) out[1 + idx] = s;
        }
    }
}
// out[0] = chi2_local, out[1 + 27 n_obj + rank] = max |diag Hcc| over the local free cameras (the other ranks' slots are
// zeroed: a SUM all-reduce then carries every rank's maximum, so the max needs no collective of its own)
// copy_to (device-resident schedule): afterwards out[0 .. copy_n) is copied there -- ALSO when the unit does not linearise: the in-place reduce of the exchange buffer
// starts from this rank's own totals every unit (a launch of its own until round 5)
__global__ __launch_bounds__(LM_THREADS) void ba_linearize_tail_kernel(const LmProblem* __restrict__ Pp, const double* __restrict__ partial, int n,
                                                                        double* __restrict__ out, int rank, int world, const double* __restrict__ ctl, int want,
                                                                        double* __restrict__ copy_to, int copy_n, double* __restrict__ fold_ctl, int* __restrict__ bad) {
    const bool live = !(ctl && (int)ctl[CTL_STATE] != want);
    if (!live) {
        if (copy_to) for (int i = threadIdx.x; i < copy_n; i += LM_THREADS) copy_to[i] = out[i];
        if (fold_ctl && threadIdx.x == 0) ba_ctl_lin(*Pp, fold_ctl, copy_to, bad);      // (clears the failure counter; the state is not "linearise": nothing else)
        return;
    }
    const LmProblem& P = *Pp;
    __shared__ double red[LM_THREADS / 64];
    double md = 0;
    for (int idx = threadIdx.x; idx < P.n_cam * 6; idx += LM_THREADS) {
        const int c = idx / 6;
        if (!P.cam_fixed[c]) md = fmax(md, fabs(P.Hcc[36 * c + DIAG21[idx - c * 6]]));
    }
    md = block_max(md, red);
    if (threadIdx.x == 0) {
        double chi = 0;
        for (int i = 0; i < n; ++i) chi += partial[i];
        out[0] = chi;
        for (int r = 0; r < world; ++r) out[1 + 27 * P.n_obj + r] = r == rank ? md : 0.0;
    }
    if (copy_to) {
        __syncthreads();                                       // thread 0's entries are written (same workgroup; the gather kernel's are from an earlier launch)
        for (int i = threadIdx.x; i < copy_n; i += LM_THREADS) copy_to[i] = out[i];
        if (fold_ctl) {
            __syncthreads();                                   // the totals are in place (one rank: they ARE the reduced ones)
            if (threadIdx.x == 0) ba_ctl_lin(P, fold_ctl, copy_to, bad);
        }
    }
}

// ---- local Schur complement for this lambda: push() + camera inverses | Y | S_g, r_g ---------------------------
__global__ __launch_bounds__(LM_THREADS) void ba_schur_cams_kernel(const LmProblem* __restrict__ Pp, double lambda, int* __restrict__ bad,
                                                                    const double* __restrict__ ctl, int want) {
    BA_GUARD(ctl, want);
    if (ctl) lambda = ctl[CTL_LAMBDA];
    const LmProblem& P = *Pp;
    for (int c = GT; c < P.n_cam; c += GS) P.cam_bak[c] = P.cam[c];
    for (int o = GT; o < P.n_obj; o += GS) P.obj_bak[o] = P.obj[o];
    // (Hcc + lambda I)^-1 and y_c = (Hcc + lambda I)^-1 b_c: an octet of lanes per camera, lane col < 6 factors the block and solves for COLUMN col of the inverse -- the
    // operations of spd_inverse6 for that column (one thread per camera ran the factorisation and six solves in a row: 7.7 us for 32 cameras); row r of y_c is summed by lane r
    // over the columns in ascending order, each taken from its lane by a shuffle: bit-identical to the one-thread form.
    for (int t = GT; t < P.n_cam * 8; t += GS) {              // GS is a multiple of 8: an octet stays together
        const int c = t >> 3, col = t & 7;
        const bool work = !P.cam_fixed[c] && col < 6;
        double x[6] = {0, 0, 0, 0, 0, 0};
        bool okc = true;
        if (work) {
            double A[36];
            unpack_sym21(P.Hcc + 36 * c, A);
            for (int d = 0; d < 6; ++d) A[d * 7] += lambda;
            okc = spd_inverse6_col(A, col, x);
            if (!okc) { for (int i = 0; i < 6; ++i) x[i] = 0; if (col == 0) atomicAdd(bad, 1); }
#pragma unroll
            for (int i = 0; i < 6; ++i) P.Hcc_inv[36 * c + i * 6 + col] = x[i];
        }
        double s = 0;
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            double xr = 0;                                      // element (row col, column k) of the inverse: lane k's x[col]
#pragma unroll
            for (int i = 0; i < 6; ++i) { const double v = __shfl(x[i], k, 8); xr = i == col ? v : xr; }
            if (work) s += xr * P.bc[6 * c + k];
        }
        if (work) P.yc[6 * c + col] = s;
    }
}
__global__ __launch_bounds__(LM_THREADS) void ba_schur_y_kernel(const LmProblem* __restrict__ Pp, const double* __restrict__ ctl, int want) {
    BA_GUARD(ctl, want);
    const LmProblem& P = *Pp;
    for (int idx = GT; idx < P.n_pair * 36; idx += GS) {
        const int p = idx / 36, rc = idx - p * 36, r = rc / 6, cc = rc - r * 6;
        const int c = P.pair_cam[p];
        double s = 0;
        if (!P.cam_fixed[c] && !P.obj_fixed[P.pair_obj[p]]) {
            const double* Hco = P.pair_part + 90 * (size_t)p + 42;
            for (int k = 0; k < 6; ++k) s += P.Hcc_inv[36 * c + r * 6 + k] * Hco[k * 6 + cc];
        }
        P.Y[idx] = s;
    }
}
// out = [S_g (ns x ns) | r_g (ns) | ok]
// One workgroup per 6 x 6 block (s1, s2) of S (grid-strided): 42 slices of object o1's camera list x the block's six rows over 252 threads (a thread keeps a row's six
// columns: one camera per slice at 32 cameras -- the three dependent index loads per camera are pure latency: one thread per element walking the whole list took 45 us,
// seven slices of 36 threads 19), slices summed in slice order through LDS (deterministic).  The right-hand side rows follow, one thread per row.
__global__ __launch_bounds__(LM_THREADS) void ba_schur_s_kernel(const LmProblem* __restrict__ Pp, int ns, const int* __restrict__ bad,
                                                                 double* __restrict__ out, const double* __restrict__ ctl, int want) {
    BA_GUARD(ctl, want);
    const LmProblem& P = *Pp;
    constexpr int NSL = LM_THREADS / 6;                       // camera slices (42 with 256 threads): six threads per slice, one per row of the block
    __shared__ double part[NSL * 36];
    __shared__ int sh_o[2];
    const int tid = threadIdx.x, nb = ns / 6;
    const int sl = tid / 6, i = tid - sl * 6;
    for (int blk = blockIdx.x; blk < nb * nb; blk += gridDim.x) {
        const int s1 = blk / nb, s2 = blk - s1 * nb;
        if (tid < 2) {
            const int want = tid == 0 ? s1 : s2;
            int oo = -1;
            for (int o = 0; o < P.n_obj; ++o) if (P.obj_slot[o] == want) oo = o;
            sh_o[tid] = oo;
        }
        __syncthreads();
        const int o1 = sh_o[0], o2 = sh_o[1];
        if (sl < NSL) {
            double acc[6] = {0, 0, 0, 0, 0, 0};
            for (int a = P.obj_pair_ptr[o1] + sl; a < P.obj_pair_ptr[o1 + 1]; a += NSL) {
                const int p1 = P.obj_pair_idx[a], c = P.pair_cam[p1];
                const int p2 = P.cam_obj_pair[(size_t)c * P.n_obj + o2];
                if (P.cam_fixed[c] || p2 < 0) continue;
                const double* H1 = P.pair_part + 90 * (size_t)p1 + 42;
                const double* Y2 = P.Y + 36 * (size_t)p2;
#pragma unroll
                for (int k = 0; k < 6; ++k) {
                    const double h = H1[k * 6 + i];
#pragma unroll
                    for (int j = 0; j < 6; ++j) acc[j] += h * Y2[k * 6 + j];
                }
            }
#pragma unroll
            for (int j = 0; j < 6; ++j) part[sl * 36 + i * 6 + j] = acc[j];
        }
        __syncthreads();
        if (tid < 36) {
            double acc = 0;
            for (int q = 0; q < NSL; ++q) acc += part[q * 36 + tid];
            out[(6 * s1 + tid / 6) * ns + 6 * s2 + tid % 6] = acc;
        }
        __syncthreads();
    }
    // the right-hand side rows: 32 lanes per row, each walking every 32nd camera of the object's list (one lane per row: 32 dependent index chains in a row, ~19 us --
    // it WAS this kernel's time), summed by an xor butterfly (the same tree on every lane)
    constexpr int RSL = 32;
    for (int t = GT; t < ns * RSL; t += GS) {
        const int row = t / RSL, rsl = t - row * RSL;
        const int s1 = row / 6, i2 = row - s1 * 6;
        int o1 = -1;
        for (int o = 0; o < P.n_obj; ++o) if (P.obj_slot[o] == s1) o1 = o;
        double acc = 0;
        for (int a = P.obj_pair_ptr[o1] + rsl; a < P.obj_pair_ptr[o1 + 1]; a += RSL) {
            const int p1 = P.obj_pair_idx[a], c = P.pair_cam[p1];
            if (P.cam_fixed[c]) continue;
            const double* H1 = P.pair_part + 90 * (size_t)p1 + 42;
            for (int k = 0; k < 6; ++k) acc += H1[k * 6 + i2] * P.yc[6 * c + k];
        }
#pragma unroll
        for (int o = 1; o < RSL; o <<= 1) acc += __shfl_xor(acc, o, 64);
        if (rsl == 0) out[ns * ns + row] = acc;
    }
    if (GT == 0) out[ns * ns + ns] = *bad == 0 ? 1.0 : 0.0;
}

// ---- solve the reduced system (identical on every rank) | back-substitute + update | chi2 | tail -----------------
// HB = [Hoo_total(21)+bo_total(6) per object], St = [S_total (ns x ns) | r_total (ns) | number of ranks whose Schur phase was ok]
// (two waves per SIMD at least = at most 256 registers per lane: the compiler then keeps the fp64 MFMA accumulators of wg_cholesky_solve in VGPRs; with 512 allowed it put
//  them in AGPRs and copied each tile out after every MFMA, behind an 18-cycle s_nop -- every MFMA completed before the next instruction issued)
__global__ __launch_bounds__(LM_THREADS, 2) void ba_solve_kernel(const LmProblem* __restrict__ Pp, double lambda, int ns,
                                                               const double* __restrict__ HB, const double* __restrict__ St, int expect_ok,
                                                               int* __restrict__ bad, const double* __restrict__ ctl, int want) {
    BA_GUARD(ctl, want);
#ifdef SUO_CHOL_PROF
    const long long t0c = clock64();
#endif
    if (ctl) lambda = ctl[CTL_LAMBDA];
    const LmProblem& P = *Pp;
    __shared__ double S[(LM_NS + 1) * (LM_NS + 1) + 8];      // odd pitch, ns + 1 rows: the right-hand side is the last (wg_cholesky_solve)
    __shared__ double rhs[LM_NS];
    __shared__ int sh_ok;
    const int tid = threadIdx.x;
    const int sp = ns | 1;                          // odd LDS pitch (lm_device.h: wg_cholesky_solve)
    const double* rt = St + ns * ns;
    if (tid == 0) sh_ok = 1;
    if (tid == 0 && expect_ok > 0 && (int)(rt[ns] + 0.5) != expect_ok) atomicAdd(bad, 1);      // some rank's camera block was singular
    // the reduced system into LDS (lower triangle only): 32 lanes per row, every load of a thread issued before its first LDS store -- ONE round trip to L2 (the buffer
    // was written by another kernel, possibly behind another XCD's L2: ~2 us a trip).  Round 5 went in batches of eight loads over the full square: five trips, 13 % of
    // this kernel.
    {
        constexpr int RG = LM_THREADS / 32, NR = (LM_NS + RG - 1) / RG, NC = (LM_NS + 31) / 32;
        const int r0 = tid >> 5, c0 = tid & 31;
        double v[NR][NC];
#pragma unroll
        for (int i = 0; i < NR; ++i)
#pragma unroll
            for (int j = 0; j < NC; ++j) {
                const int row = r0 + i * RG, col = c0 + 32 * j;
                v[i][j] = St[(row < ns && col <= row) ? row * ns + col : 0];      // (an address whatever the lane: no predicated loads, no branches between them)
            }
#pragma unroll
        for (int i = 0; i < NR; ++i)
#pragma unroll
            for (int j = 0; j < NC; ++j) {
                const int row = r0 + i * RG, col = c0 + 32 * j;
                if (row < ns && col <= row) S[row * sp + col] = -v[i][j];
            }
    }
    __syncthreads();
    for (int idx = tid; idx < P.n_obj * 36; idx += LM_THREADS) {
        const int o = idx / 36, rc = idx - o * 36, r = rc / 6, cc = rc - r * 6;
        const int so = P.obj_slot[o];
        if (so < 0) continue;
        const int rr = r < cc ? r : cc, c2 = r < cc ? cc : r;
        const int packed = rr * 6 - rr * (rr - 1) / 2 + (c2 - rr);
        S[(6 * so + r) * sp + 6 * so + cc] += HB[27 * o + packed] + (r == cc ? lambda : 0.0);
    }
    for (int idx = tid; idx < P.n_obj * 6; idx += LM_THREADS) {
        const int o = idx / 6, r = idx - o * 6;
        if (P.obj_slot[o] >= 0) rhs[6 * P.obj_slot[o] + r] = HB[27 * o + 21 + r] - rt[6 * P.obj_slot[o] + r];
    }
    __syncthreads();
#ifdef SUO_CHOL_PROF
    long long pt[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    const long long t1 = clock64();
    wg_cholesky_solve_mfma<LM_THREADS>(S, sp, rhs, ns, tid, &sh_ok, pt);
    __syncthreads();
    { static __device__ int calls = 0; if (tid == 0) { int c = atomicAdd(&calls, 1); if (c % 50 == 20) printf("ba_solve ns=%d: load %lld | rhs copy + tiles into registers %lld  diag %lld  panel %lld  barrier %lld  store diag + fragment loads %lld  mfma %lld  column write-back %lld  barrier %lld  backward %lld  (thread 0)\n", ns, t1 - t0c, pt[0], pt[1], pt[2], pt[3], pt[8], pt[4], pt[5], pt[6], pt[7]); } }
#else
    wg_cholesky_solve_mfma<LM_THREADS>(S, sp, rhs, ns, tid, &sh_ok);                  // blocked by 6, the whole workgroup (lm_device.h)
    __syncthreads();
#endif
    for (int idx = tid; idx < P.n_obj * 6; idx += LM_THREADS) {
        const int o = idx / 6;
        P.xo[idx] = P.obj_slot[o] >= 0 ? rhs[6 * P.obj_slot[o] + (idx - o * 6)] : 0.0;
    }
    if (tid == 0 && sh_ok == 0) atomicAdd(bad, 1);
}
// The same solve for reduced systems beyond LM_NS rows (more than 16 free objects: T-LESS scenes; the reference's graph
// has no such limit, lib/object_slam.py:746-778): S and the right-hand side live in a global scratch buffer (L2-resident,
// ns^2 doubles), factorised by the whole workgroup with a right-looking Cholesky -- one barrier triple per column.  Only the
// summation order differs from the LDS path.
__global__ __launch_bounds__(LM_THREADS) void ba_solve_big_kernel(const LmProblem* __restrict__ Pp, double lambda, int ns,
                                                                   const double* __restrict__ HB, const double* __restrict__ St, int expect_ok,
                                                                   int* __restrict__ bad, double* __restrict__ S, double* __restrict__ rhs,
                                                                   const double* __restrict__ ctl, int want) {
    BA_GUARD(ctl, want);
    if (ctl) lambda = ctl[CTL_LAMBDA];
    const LmProblem& P = *Pp;
    __shared__ int sh_ok;
    __shared__ double sh_d;
    const int tid = threadIdx.x;
    const double* rt = St + ns * ns;
    if (tid == 0) sh_ok = 1;
    if (tid == 0 && expect_ok > 0 && (int)(rt[ns] + 0.5) != expect_ok) atomicAdd(bad, 1);
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
    for (int j = 0; j < ns; ++j) {                                  // S = L L^T, lower triangle in place
        if (tid == 0) {
            const double piv = S[j * ns + j];
            if (!(piv > 0) || !isfinite(piv)) sh_ok = 0;
            sh_d = sqrt(piv > 0 ? piv : 1.0);
            S[j * ns + j] = sh_d;
        }
        __syncthreads();
        const double d = sh_d;
        for (int i = j + 1 + tid; i < ns; i += LM_THREADS) S[i * ns + j] /= d;
        __syncthreads();
        const int m = ns - j - 1;                                    // trailing update of the lower triangle, rows j+1 .. ns-1
        for (int idx = tid; idx < m * m; idx += LM_THREADS) {
            const int i = j + 1 + idx / m, k = j + 1 + idx % m;
            if (k <= i) S[i * ns + k] -= S[i * ns + j] * S[k * ns + j];
        }
        __syncthreads();
    }
    for (int j = 0; j < ns; ++j) {                                  // L y = rhs
        if (tid == 0) rhs[j] /= S[j * ns + j];
        __syncthreads();
        const double yj = rhs[j];
        for (int i = j + 1 + tid; i < ns; i += LM_THREADS) rhs[i] -= S[i * ns + j] * yj;
        __syncthreads();
    }
    for (int j = ns - 1; j >= 0; --j) {                             // L^T x = y
        if (tid == 0) rhs[j] /= S[j * ns + j];
        __syncthreads();
        const double xj = rhs[j];
        for (int i = tid; i < j; i += LM_THREADS) rhs[i] -= S[j * ns + i] * xj;
        __syncthreads();
    }
    for (int idx = tid; idx < P.n_obj * 6; idx += LM_THREADS) {
        const int o = idx / 6;
        P.xo[idx] = P.obj_slot[o] >= 0 ? rhs[6 * P.obj_slot[o] + (idx - o * 6)] : 0.0;
    }
    if (tid == 0 && sh_ok == 0) atomicAdd(bad, 1);
}
// x_c = y_c - sum_o Y(c,o) x_o for the own cameras, then T <- exp(x) T for cameras and objects.  24 threads per camera: the six rows of x_c times four slices of the
// camera's pair list (its dependent index chain is pure latency: one thread per camera took 47 us for 32 cameras, six 14), the slices summed by an xor butterfly.
__global__ __launch_bounds__(LM_THREADS) void ba_update_kernel(const LmProblem* __restrict__ Pp, const int* __restrict__ bad, const double* __restrict__ ctl, int want) {
    BA_GUARD(ctl, want);
    const LmProblem& P = *Pp;
    const bool ok = *bad == 0;
    constexpr int SLU = 4, TPC = 6 * SLU, CPW = LM_THREADS / TPC;      // cameras per workgroup and sweep
    const int tid = threadIdx.x, cl = tid / TPC, rs = tid - cl * TPC, r = rs / SLU, sl = rs - r * SLU;
    for (int c0 = blockIdx.x * CPW; c0 < P.n_cam; c0 += gridDim.x * CPW) {
        const int c = c0 + cl;
        const bool mine = cl < CPW && c < P.n_cam;
        const bool free_cam = mine && !P.cam_fixed[c];
        double s = 0;
        if (free_cam) {
            for (int b = P.cam_pair_ptr[c] + sl; b < P.cam_pair_ptr[c + 1]; b += SLU) {
                const int p = P.cam_pair_idx[b], o = P.pair_obj[p];
                if (P.obj_fixed[o]) continue;
                const double* Yr = P.Y + 36 * (size_t)p + r * 6;
                const double* xo = P.xo + 6 * o;
                for (int k = 0; k < 6; ++k) s -= Yr[k] * xo[k];
            }
        }
        s += __shfl_xor(s, 1, 64);
        s += __shfl_xor(s, 2, 64);
        if (mine && sl == 0) P.xc[6 * c + r] = free_cam ? P.yc[6 * c + r] + s : 0.0;
        __syncthreads();                                      // the six rows of a camera are written (same workgroup)
        if (free_cam && rs == 0 && ok) pose_oplus(P.cam[c], P.xc + 6 * c);
    }
    // the objects' updates by the LAST threads of the grid: the first ones have just walked a camera each (another exponential map in the same thread's way)
    if (ok)
        for (int o = GS - 1 - GT; o < P.n_obj; o += GS) if (!P.obj_fixed[o]) pose_oplus(P.obj[o], P.xo + 6 * o);
}
// out = [chi2_local after the step | sum x_c (lambda x_c + b_c) over own cameras | ok | the same sum over the objects]
// (the first three are summed over ranks; the object part is identical on every rank)
__global__ __launch_bounds__(LM_THREADS) void ba_update_tail_kernel(const LmProblem* __restrict__ Pp, double lambda, const double* __restrict__ partial,
                                                                     int n, const double* __restrict__ HB, const int* __restrict__ bad,
                                                                     double* __restrict__ out, const double* __restrict__ ctl, int want, double* __restrict__ fold_ctl) {
    BA_GUARD(ctl, want);
    if (ctl) lambda = ctl[CTL_LAMBDA];
    const LmProblem& P = *Pp;
    __shared__ double red[LM_THREADS / 64];
    const int tid = threadIdx.x;
    const bool ok = *bad == 0;
    double sc_c = 0, sc_o = 0;
    if (ok) {
        for (int idx = tid; idx < P.n_cam * 6; idx += LM_THREADS)
            if (!P.cam_fixed[idx / 6]) sc_c += P.xc[idx] * (lambda * P.xc[idx] + P.bc[idx]);
        for (int idx = tid; idx < P.n_obj * 6; idx += LM_THREADS)
            if (!P.obj_fixed[idx / 6]) sc_o += P.xo[idx] * (lambda * P.xo[idx] + HB[27 * (idx / 6) + 21 + (idx % 6)]);
    }
    sc_c = block_sum(sc_c, red);
    sc_o = block_sum(sc_o, red);
    if (tid == 0) {
        double chi = 0;
        for (int i = 0; i < n; ++i) chi += partial[i];
        out[0] = chi; out[1] = sc_c; out[2] = ok ? 1.0 : 0.0; out[3] = sc_o;
    }
    if (fold_ctl) {                                            // one rank: `out` is the reduced result -- decide, and pop() a rejected step, right here
        __shared__ int sh_restore;
        if (tid == 0) { ba_decide(fold_ctl, out); sh_restore = fold_ctl[CTL_RESTORE] != 0.0; }
        __syncthreads();
        if (sh_restore) {
            for (int c = tid; c < P.n_cam; c += LM_THREADS) P.cam[c] = P.cam_bak[c];
            for (int o = tid; o < P.n_obj; o += LM_THREADS) P.obj[o] = P.obj_bak[o];
        }
    }
}

// ---- pop(): restore the poses saved by the Schur phase -------------------------------------------------------
__global__ __launch_bounds__(LM_THREADS) void ba_restore_kernel(const LmProblem* __restrict__ Pp, const double* __restrict__ ctl) {
    if (ctl && ctl[CTL_RESTORE] == 0.0) return;
    const LmProblem& P = *Pp;
    for (int c = GT; c < P.n_cam; c += GS) P.cam[c] = P.cam_bak[c];
    for (int o = GT; o < P.n_obj; o += GS) P.obj[o] = P.obj_bak[o];
}

__global__ __launch_bounds__(LM_THREADS) void ba_finalize_kernel(const LmProblem* __restrict__ Pp) {
    const LmProblem& P = *Pp;
    for (int c = GT; c < P.n_cam; c += GS) pose_to_T(P.cam[c], P.cam_T + 12 * c);
    for (int o = GT; o < P.n_obj; o += GS) pose_to_T(P.obj[o], P.obj_T + 12 * o);
}

// ---- the schedule's own steps (one thread each: a dozen scalar operations) ---------------------------------------------------------
// start of a round (ObjectSLAM.optimize's optimizer.optimize(n), object_slam.py:873-875): n iterations to go, the next unit linearises
__global__ void ba_ctl_begin_kernel(double* __restrict__ ctl, int its, int world) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        ctl[CTL_STATE] = its > 0 ? ST_LINEARIZE : ST_DONE;
        ctl[CTL_IT] = 0; ctl[CTL_ITS] = its; ctl[CTL_QMAX] = 0; ctl[CTL_RHO] = 0; ctl[CTL_RESTORE] = 0; ctl[CTL_WORLD] = world;
        ctl[CTL_LAMBDA] = -1; ctl[CTL_NI] = 2;
    }
}
// after the linearisation totals are reduced: chi2 of the iteration; in the first iteration computeLambdaInit (tau * max |diag H| over ALL
// free vertices: the cameras' maxima travel in per-rank slots of the same SUM, the objects' diagonals are in the totals)
__device__ void ba_ctl_lin(const LmProblem& P, double* __restrict__ ctl, const double* __restrict__ lin, int* __restrict__ bad) {
    if (bad) *bad = 0;                                         // the Schur phase's failure counter (a memset node of its own until round 5)
    if ((int)ctl[CTL_STATE] != ST_LINEARIZE) return;
    ctl[CTL_CHI] = lin[0];
    if ((int)ctl[CTL_IT] == 0) {
        double maxd = 0;
        const int world = (int)ctl[CTL_WORLD];
        for (int r = 0; r < world; ++r) maxd = fmax(maxd, lin[1 + 27 * P.n_obj + r]);
        for (int o = 0; o < P.n_obj; ++o)
            if (!P.obj_fixed[o])
                for (int d = 0; d < 6; ++d) maxd = fmax(maxd, fabs(lin[1 + 27 * o + DIAG21[d]]));
        ctl[CTL_LAMBDA] = 1e-5 * maxd;
        ctl[CTL_NI] = 2;
    }
    ctl[CTL_QMAX] = 0; ctl[CTL_RHO] = 0;
    ctl[CTL_STATE] = ST_TRIAL;
}
__global__ void ba_ctl_lin_kernel(const LmProblem* __restrict__ Pp, double* __restrict__ ctl, const double* __restrict__ lin, int* __restrict__ bad) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    ba_ctl_lin(*Pp, ctl, lin, bad);
}
// after the step's [chi2 | scale over cameras | ok-count] are reduced (red[3] = scale over objects, identical on every rank): the gain ratio,
// accept / reject, lambda / ni, the trial and iteration counters, what the next unit is
// ... and pop() when the trial was rejected: one workgroup restores the poses saved by the Schur phase behind thread 0's decision (a grid-strided launch of its own
// until round 5; a rank's few thousand poses are a few microseconds for 256 threads)
__device__ void ba_decide(double* __restrict__ ctl, const double* __restrict__ red) {
    ctl[CTL_RESTORE] = 0;
    if ((int)ctl[CTL_STATE] != ST_TRIAL) return;
    const int world = (int)ctl[CTL_WORLD];
    double lambda = ctl[CTL_LAMBDA], ni = ctl[CTL_NI], current = ctl[CTL_CHI];
    double temp = 1.7976931348623157e308, scale = 0;
    if ((int)(red[2] + 0.5) == world) { temp = red[0]; scale = red[1] + red[3]; }
    const double rho = (current - temp) / (scale + 1e-3);
    bool lam_finite = true;
    int qmax = (int)ctl[CTL_QMAX];
    if (rho > 0 && isfinite(temp)) {
        const double r21 = 2 * rho - 1;
        double alpha = 1. - r21 * r21 * r21;
        alpha = fmin(alpha, 2. / 3.);
        lambda *= fmax(1. / 3., alpha);
        ni = 2;
        current = temp;
    } else {
        lambda *= ni;
        ni *= 2;
        ctl[CTL_RESTORE] = 1;                                  // pop()
        if (!isfinite(lambda)) lam_finite = false;
    }
    if (lam_finite) { ++qmax; ctl[CTL_TRIALS] += 1; }
    ctl[CTL_LAMBDA] = lambda; ctl[CTL_NI] = ni; ctl[CTL_CHI] = current; ctl[CTL_RHO] = rho; ctl[CTL_QMAX] = qmax;
    if (lam_finite && rho < 0 && qmax < 10) return;            // another trial on the same linearisation
    ctl[CTL_LM_ITS] += 1;
    const int it = (int)ctl[CTL_IT] + 1;
    ctl[CTL_IT] = it;
    ctl[CTL_STATE] = (qmax == 10 || rho == 0 || !lam_finite || it >= (int)ctl[CTL_ITS]) ? ST_DONE : ST_LINEARIZE;
}
__global__ __launch_bounds__(LM_THREADS) void ba_ctl_decide_kernel(const LmProblem* __restrict__ Pp, double* __restrict__ ctl, const double* __restrict__ red) {
    __shared__ int sh_restore;
    if (threadIdx.x == 0) { ba_decide(ctl, red); sh_restore = ctl[CTL_RESTORE] != 0.0; }
    __syncthreads();
    if (!sh_restore) return;
    const LmProblem& P = *Pp;
    for (int c = threadIdx.x; c < P.n_cam; c += LM_THREADS) P.cam[c] = P.cam_bak[c];
    for (int o = threadIdx.x; o < P.n_obj; o += LM_THREADS) P.obj[o] = P.obj_bak[o];
}
__global__ void ba_copy_kernel(const double* __restrict__ src, double* __restrict__ dst, int n) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) dst[i] = src[i];
}

#define BA_GRID(k, ...) hipLaunchKernelGGL(k, dim3(BA_WGS), dim3(LM_THREADS), 0, s, (const LmProblem*)P, ##__VA_ARGS__)
#define BA_ONE(k, ...) hipLaunchKernelGGL(k, dim3(1), dim3(LM_THREADS), 0, s, (const LmProblem*)P, ##__VA_ARGS__)
#define BA_DONE SUO_HIP_CHECK(hipGetLastError()); return SUO_OK;

size_t ba_scratch_doubles() { return BA_SCRATCH_DOUBLES; }

int launch_ba_init(const void* P, hipStream_t s) { BA_GRID(ba_init_kernel); BA_DONE }
int launch_ba_classify(const void* P, int keep_all, double* out, double* scratch, hipStream_t s) {
    BA_GRID(ba_classify_kernel, keep_all, scratch);
    hipLaunchKernelGGL(ba_sum_kernel, dim3(1), dim3(64), 0, s, (const double*)scratch, BA_WGS, out, 0);
    BA_DONE
}
int launch_ba_linearize(const void* P, int robust_on, double* out, double* scratch, int rank, int world, hipStream_t s, const double* ctl, double* copy_to, int copy_n,
                        double* fold_ctl) {
    BA_GRID(ba_edge_pass_kernel, robust_on, 1, scratch, ctl, ST_LINEARIZE);
    // (one (pair, entry) item per thread up to 256 workgroups: with the 64 of the other steps a thread walked three items' edge lists one after the other)
    hipLaunchKernelGGL(ba_accumulate_kernel, dim3(BA_WGS * 4), dim3(LM_THREADS), 0, s, (const LmProblem*)P, ctl, ST_LINEARIZE);
    BA_GRID(ba_gather_kernel, out, ctl, ST_LINEARIZE);
    BA_ONE(ba_linearize_tail_kernel, (const double*)scratch, BA_WGS, out, rank, world, ctl, ST_LINEARIZE, copy_to, copy_n, fold_ctl, (int*)(scratch + BA_WGS));
    BA_DONE
}
int launch_ba_schur(const void* P, double lambda, int ns, double* out, double* scratch, hipStream_t s, const double* ctl) {
    int* bad = (int*)(scratch + BA_WGS);
    if (!ctl) SUO_HIP_CHECK(hipMemsetAsync(bad, 0, sizeof(int), s));      // (device-resident schedule: ba_ctl_lin_kernel has cleared it)
    BA_GRID(ba_schur_cams_kernel, lambda, bad, ctl, ST_TRIAL);
    BA_GRID(ba_schur_y_kernel, ctl, ST_TRIAL);
    {   // one workgroup per 6 x 6 block of S
        const int nb = ns / 6, g = nb * nb < BA_WGS ? BA_WGS : (nb * nb > 1024 ? 1024 : nb * nb);
        hipLaunchKernelGGL(ba_schur_s_kernel, dim3(g), dim3(LM_THREADS), 0, s, (const LmProblem*)P, ns, (const int*)bad, out, ctl, ST_TRIAL);
    }
    BA_DONE
}
int launch_ba_solve_update(const void* P, double lambda, int ns, int robust_on, const double* HB, const double* St, int expect_ok, double* out,
                           double* scratch, double* big, hipStream_t s, const double* ctl, double* fold_ctl) {
    int* bad = (int*)(scratch + BA_WGS);          // carries over from the Schur phase of the same trial
    if (ns <= LM_NS) BA_ONE(ba_solve_kernel, lambda, ns, HB, St, expect_ok, bad, ctl, ST_TRIAL);
    else if (!big) { suo_set_error("bundle adjustment: reduced system of %d rows needs the big-system scratch", ns); return SUO_ERR_ARG; }
    else BA_ONE(ba_solve_big_kernel, lambda, ns, HB, St, expect_ok, bad, big, big + (size_t)ns * ns, ctl, ST_TRIAL);
    BA_GRID(ba_update_kernel, (const int*)bad, ctl, ST_TRIAL);
    BA_GRID(ba_edge_pass_kernel, robust_on, 0, scratch, ctl, ST_TRIAL);
    BA_ONE(ba_update_tail_kernel, lambda, (const double*)scratch, BA_WGS, HB, (const int*)bad, out, ctl, ST_TRIAL, fold_ctl);
    BA_DONE
}
// test entry (suo_debug_cholesky_solve): wg_cholesky_solve on a dense symmetric system, as ba_solve_kernel calls it
__global__ __launch_bounds__(LM_THREADS, 2) void debug_cholesky_kernel(const double* __restrict__ A, const double* __restrict__ b, int ns, double* __restrict__ x, int* __restrict__ ok_out) {
    __shared__ double S[(LM_NS + 1) * (LM_NS + 1) + 8];
    __shared__ double rhs[LM_NS];
    __shared__ int sh_ok;
    const int tid = threadIdx.x, sp = ns | 1;
    if (tid == 0) sh_ok = 1;
    for (int idx = tid; idx < ns * ns; idx += LM_THREADS) { const int row = idx / ns, col = idx - row * ns; if (col <= row) S[row * sp + col] = A[idx]; }
    for (int i = tid; i < ns; i += LM_THREADS) rhs[i] = b[i];
    __syncthreads();
#ifdef SUO_CHOL_PROF
    long long ptd[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    wg_cholesky_solve_mfma<LM_THREADS>(S, sp, rhs, ns, tid, &sh_ok, ptd);
#else
    wg_cholesky_solve_mfma<LM_THREADS>(S, sp, rhs, ns, tid, &sh_ok);
#endif
    __syncthreads();
    for (int i = tid; i < ns; i += LM_THREADS) x[i] = rhs[i];
    if (tid == 0) *ok_out = sh_ok;
}
int launch_debug_cholesky(const double* A, const double* b, int ns, double* x, int* ok, hipStream_t s) {
    if (ns <= 0 || ns > LM_NS || ns % 6) { suo_set_error("suo_debug_cholesky_solve: ns = %d (a multiple of 6, at most %d)", ns, LM_NS); return SUO_ERR_ARG; }
    hipLaunchKernelGGL(debug_cholesky_kernel, dim3(1), dim3(LM_THREADS), 0, s, A, b, ns, x, ok);
    SUO_HIP_CHECK(hipGetLastError());
    return SUO_OK;
}
int launch_ba_ctl_begin(double* ctl, int its, int world, hipStream_t s) {
    hipLaunchKernelGGL(ba_ctl_begin_kernel, dim3(1), dim3(64), 0, s, ctl, its, world);
    SUO_HIP_CHECK(hipGetLastError());
    return SUO_OK;
}
int launch_ba_ctl_lin(const void* P, double* ctl, const double* lin, double* scratch, hipStream_t s) {
    hipLaunchKernelGGL(ba_ctl_lin_kernel, dim3(1), dim3(64), 0, s, (const LmProblem*)P, ctl, lin, (int*)(scratch + BA_WGS));
    SUO_HIP_CHECK(hipGetLastError());
    return SUO_OK;
}
int launch_ba_ctl_decide(const void* P, double* ctl, const double* red, hipStream_t s) {
    BA_ONE(ba_ctl_decide_kernel, ctl, red);                    // the decision + pop() when the trial was rejected
    BA_DONE
}
int launch_ba_copy(const double* src, double* dst, int n, hipStream_t s) {
    hipLaunchKernelGGL(ba_copy_kernel, dim3((n + 255) / 256 > 64 ? 64 : (n + 255) / 256), dim3(256), 0, s, src, dst, n);
    SUO_HIP_CHECK(hipGetLastError());
    return SUO_OK;
}
int launch_ba_restore(const void* P, hipStream_t s) { BA_GRID(ba_restore_kernel, (const double*)nullptr); BA_DONE }
int launch_ba_finalize(const void* P, hipStream_t s) { BA_GRID(ba_finalize_kernel); BA_DONE }

}  // namespace suo
