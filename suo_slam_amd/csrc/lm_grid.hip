// The LM / bundle-adjustment algorithm of csrc/lm.hip for ONE large graph spread over many workgroups.
//
// The global SLAM adjustment (every 10 views: all cameras x objects x keypoints, /root/reference/lib/object_slam.py:
// 444-447,703-903) has thousands of edges; one workgroup on one CU spends 150 ms on 60 cameras / 7500 edges, almost all
// of it in per-edge / per-pair / per-camera loops that are embarrassingly parallel.  Here G workgroups (one per CU of a
// single XCD, so they share an L2) execute the same rounds / iterations / trials in lock-step: every loop is grid-strided,
// every workgroup barrier that orders data between phases is a grid barrier, every reduction is summed in workgroup
// order (deterministic), and the (<= 96 x 96) reduced system is factorised redundantly by every workgroup.  All state lives in the
// problem's HBM arrays (no LDS relocation); decisions (gain ratio, lambda, termination) are taken redundantly by every
// workgroup from identical reduced values, so control flow never diverges between workgroups.
//
// Same arithmetic per edge / block entry as csrc/lm.hip; only the partition of the chi2 sums differs (rounding-level).
#include "lm_device.h"

namespace suo {

// -DSUO_LG_PROFILE: thread 0 of workgroup 0 charges the wall clock between consecutive LGPROF(i) marks to section i
// (read back with suo_debug_lg_prof; tools/bench_global_ba.py prints it when the symbol exists)
#ifdef SUO_LG_PROFILE
__device__ long long g_lg_prof[16];
#define LGPROF(i) do { if (gt == 0) { const long long _t = wall_clock64(); g_lg_prof[(i)] += _t - lgprof_t; lgprof_t = _t; } } while (0)
#else
#define LGPROF(i) do { } while (0)
#endif

constexpr int LG_THREADS = 256;
constexpr int LG_MAX_WGS = 32;                // CUs of one XCD
constexpr int LG_RED_N = 3;                   // values one grid reduction can carry
constexpr int LG_IDX_MAX = 4096;              // pairs whose index arrays are mirrored in LDS (2 x 16 KB)
constexpr int LG_TAB_MAX = 8192;              // entries of the camera x object pair table mirrored in LDS (32 KB)

struct LmGridScratch {
    unsigned* bar;            // [2]: arrivals, generation (zeroed by the host before the launch)
    double* red;              // [2][LG_RED_N][LG_MAX_WGS] alternating reduction buffers
    double* S;                // [ns*ns + ns] reduced system assembled by the whole grid, factorised by every workgroup
};

struct GridCtx {
    unsigned* bar; double* red; int G, wg; unsigned red_cnt;
    bool same_xcd;            // every cooperating workgroup reported the same XCC id: they share one L2
};

// Grid barrier.  General form: agent-scope release / acquire fences, which on this multi-die part write back and invalidate
// L2 (buffer_wbl2 sc1 / buffer_inv sc1) -- ~45 us per barrier, i.e. most of an LM trial.  When all cooperating workgroups
// sit on ONE XCD (verified at kernel start from the XCC_ID hardware register, not assumed) they share a single L2, so it is
// enough that each wave's stores have reached L2 (s_waitcnt vmcnt(0): the vector L1 is write-through) before arriving and
// that the CU's L1 is dropped (buffer_inv sc1, by one wave per workgroup) after leaving; the barrier words are only touched by L2 atomics.
DEV void grid_sync(GridCtx& g) {
    if (g.same_xcd) {
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __syncthreads();
        if (threadIdx.x == 0) {
            const unsigned gen = __hip_atomic_fetch_add(&g.bar[1], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (__hip_atomic_fetch_add(&g.bar[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (unsigned)g.G - 1) {
                __hip_atomic_exchange(&g.bar[0], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __hip_atomic_fetch_add(&g.bar[1], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            } else {
                while (__hip_atomic_fetch_add(&g.bar[1], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gen) __builtin_amdgcn_s_sleep(1);
            }
        }
        __syncthreads();
        // the CU's vector L1 is dropped by ONE wave for all (the cache is the CU's, not the wave's): `buffer_inv sc1` from every wave of every workgroup costs
        // ~15 us per barrier on this multi-XCD part, from one wave per workgroup ~1 us (measured with the barrier probe of commit 98cd5bc, profiles/REJECTED.md: 16.6 / 3.2 / 2.2 us per barrier with all waves /
        // one wave / no invalidate) -- 1.3 of the 11.9 ms of a global SLAM adjustment
        if (threadIdx.x < 64) asm volatile("buffer_inv sc1\n\ts_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        return;
    }
    __threadfence();                                                         // release: every wave's writes -> L2 -> memory
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned gen = __hip_atomic_load(&g.bar[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (__hip_atomic_fetch_add(&g.bar[0], 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) == (unsigned)g.G - 1) {
            __hip_atomic_store(&g.bar[0], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_fetch_add(&g.bar[1], 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            while (__hip_atomic_load(&g.bar[1], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) == gen) __builtin_amdgcn_s_sleep(1);
        }
    }
    __syncthreads();
    __threadfence();                                                         // acquire: drop stale cache lines (every wave)
}

// deterministic grid reductions: workgroup partial (fixed wave order) -> red[buf][wg] -> all sum / max in workgroup order
DEV double grid_reduce(double v, bool is_max, GridCtx& g, double* red_lds) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { const double u = __shfl_xor(v, o, 64); v = is_max ? fmax(v, u) : v + u; }
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red_lds[threadIdx.x >> 6] = v;
    __syncthreads();
    double* buf = g.red + (g.red_cnt++ & 1) * LG_RED_N * LG_MAX_WGS;
    if (threadIdx.x == 0) {
        double s = red_lds[0];
        for (int i = 1; i < LG_THREADS / 64; ++i) s = is_max ? fmax(s, red_lds[i]) : s + red_lds[i];
        buf[g.wg] = s;
    }
    grid_sync(g);
    double s = buf[0];
    for (int i = 1; i < g.G; ++i) s = is_max ? fmax(s, buf[i]) : s + buf[i];
    return s;
}
// three values behind ONE grid barrier: v[0], v[1] summed, v[2] maximised (same orders as grid_reduce)
DEV void grid_reduce3(double (&v)[LG_RED_N], GridCtx& g, double* red_lds) {
#pragma unroll
    for (int k = 0; k < LG_RED_N; ++k)
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { const double u = __shfl_xor(v[k], o, 64); v[k] = k == 2 ? fmax(v[k], u) : v[k] + u; }
    __syncthreads();
    if ((threadIdx.x & 63) == 0)
        for (int k = 0; k < LG_RED_N; ++k) red_lds[k * (LG_THREADS / 64) + (threadIdx.x >> 6)] = v[k];
    __syncthreads();
    double* buf = g.red + (g.red_cnt++ & 1) * LG_RED_N * LG_MAX_WGS;
    if (threadIdx.x < LG_RED_N) {
        const int k = threadIdx.x;
        double s = red_lds[k * (LG_THREADS / 64)];
        for (int i = 1; i < LG_THREADS / 64; ++i) s = k == 2 ? fmax(s, red_lds[k * (LG_THREADS / 64) + i]) : s + red_lds[k * (LG_THREADS / 64) + i];
        buf[k * LG_MAX_WGS + g.wg] = s;
    }
    grid_sync(g);
#pragma unroll
    for (int k = 0; k < LG_RED_N; ++k) {
        double s = buf[k * LG_MAX_WGS];
        for (int i = 1; i < g.G; ++i) s = k == 2 ? fmax(s, buf[k * LG_MAX_WGS + i]) : s + buf[k * LG_MAX_WGS + i];
        v[k] = s;
    }
}

__global__ __launch_bounds__(LG_THREADS) void lm_grid_kernel(const LmProblem* __restrict__ Pp, LmGridScratch sc, int G) {
    if (blockIdx.x & 7) return;                 // workgroup b runs on XCD b % 8 (observed; performance only): keep one XCD
    const LmProblem& P = *Pp;
    __shared__ double S[LM_NS * (LM_NS + 1) + 8];      // odd pitch (+ slack)
    __shared__ double rhs[LM_NS];
    __shared__ double red_lds[LG_RED_N * (LG_THREADS / 64)];
    __shared__ int sh_flag, sh_ok;
    GridCtx g;
    g.bar = sc.bar; g.red = sc.red; g.G = G; g.wg = blockIdx.x >> 3; g.red_cnt = 0; g.same_xcd = false;
    const int tid = threadIdx.x;
    {   // which XCD is every cooperating workgroup on?  (HW_REG_XCC_ID = 20, bits [3:0]); one general barrier, then all agree
        if (tid == 0) sc.red[g.wg] = (double)(__builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11)) & 15);
        grid_sync(g);
        bool same = true;
        for (int i = 1; i < G; ++i) same = same && (sc.red[i] == sc.red[0]);
        grid_sync(g);                           // everyone has read the ids before the buffer is reused by reductions
        g.same_xcd = same;
    }
    const int gt = g.wg * LG_THREADS + tid, GS = G * LG_THREADS;      // position / stride of the grid-strided loops
#ifdef SUO_LG_PROFILE
    long long lgprof_t = wall_clock64();
#endif

    if (tid == 0) {
        int ns = 0, nfc = 0;
        for (int o = 0; o < P.n_obj; ++o) { if (g.wg == 0) P.obj_slot[o] = P.obj_fixed[o] ? -1 : ns; if (!P.obj_fixed[o]) ++ns; }
        for (int c = 0; c < P.n_cam; ++c) nfc += P.cam_fixed[c] ? 0 : 1;
        sh_flag = ns | (nfc << 16);
    }
    __syncthreads();
    const int n_free_obj = sh_flag & 0xffff, n_free_cam = sh_flag >> 16;
    const bool schur = n_free_cam > 0 && n_free_obj > 0;
    const int ns = 6 * n_free_obj;
    if (schur && n_free_obj > LM_MAX_SCHUR_OBJ) {
        if (gt == 0) { P.stats[0] = -1; P.stats[1] = P.stats[2] = P.stats[3] = 0; }
        return;
    }
    // LDS copies of the small index arrays the Schur sums chase (global pointers stay in use when a graph is too large)
    __shared__ int l_slot_obj[LM_NS / 6 + 1];
    __shared__ int l_pair_cam[LG_IDX_MAX], l_obj_pair_idx[LG_IDX_MAX], l_cam_obj[LG_TAB_MAX];
    const bool idx_in_lds = P.n_pair <= LG_IDX_MAX && (long)P.n_cam * P.n_obj <= LG_TAB_MAX;
    if (idx_in_lds) {
        for (int k = tid; k < P.n_pair; k += LG_THREADS) { l_pair_cam[k] = P.pair_cam[k]; l_obj_pair_idx[k] = P.obj_pair_idx[k]; }
        for (int k = tid; k < P.n_cam * P.n_obj; k += LG_THREADS) l_cam_obj[k] = P.cam_obj_pair[k];
    }
    if (tid == 0) { int s = 0; for (int o = 0; o < P.n_obj; ++o) if (!P.obj_fixed[o] && s <= LM_NS / 6) l_slot_obj[s++] = o; }
    __syncthreads();
    const int* x_pair_cam = idx_in_lds ? l_pair_cam : P.pair_cam;
    const int* x_obj_pair_idx = idx_in_lds ? l_obj_pair_idx : P.obj_pair_idx;
    const int* x_cam_obj = idx_in_lds ? l_cam_obj : P.cam_obj_pair;
    for (int c = gt; c < P.n_cam; c += GS) pose_from_T(P.cam_T + 12 * c, P.cam[c]);
    for (int o = gt; o < P.n_obj; o += GS) pose_from_T(P.obj_T + 12 * o, P.obj[o]);
    for (int e = gt; e < P.n_edge; e += GS) P.level[e] = 0;
    grid_sync(g);

    // ---- initial classification (object_slam.py:848-866) --------------------------------------
    int num_good;
    if (P.init_with_outliers) {
        num_good = P.n_edge;
    } else {
        double my_good = 0;
        for (int e = gt; e < P.n_edge; e += GS) {
            double er[2];
            edge_error(P, e, er, nullptr, nullptr);
            const double c2 = edge_chi2(P, e, er);
            P.edge_chi2[e] = c2;
            if (c2 > P.chi2_thr) { P.level[e] = 1; P.edge_inlier[e] = 0; }
            else { P.level[e] = 0; P.edge_inlier[e] = 1; my_good += 1; }
        }
        num_good = (int)grid_reduce(my_good, false, g, red_lds);
    }
    bool robust_on = true;
    int rounds = 0, lm_its = 0, lm_trials = 0;
    const int drop = (P.n_rounds / 2) > 1 ? (P.n_rounds / 2) : 1;
    const int diag21[6] = {0, 6, 11, 15, 18, 20};

    for (int round = 0; round < P.n_rounds; ++round) {
        if (P.n_edge < 4 || num_good < 4) break;
        ++rounds;
        double nact = 0;
        for (int e = gt; e < P.n_edge; e += GS) nact += edge_active(P, e) ? 1.0 : 0.0;
        nact = grid_reduce(nact, false, g, red_lds);
        const int iterations = nact > 0 ? P.its[round] : 0;
        double lambda = -1, ni = 2;
        for (int it = 0; it < iterations; ++it) {
            // ---- errors, chi2, Jacobians (HBM), pair blocks, diagonal blocks ---------------------
            LGPROF(9);
            double currentChi = grid_reduce(edge_pass_partial(P, 0, P.n_edge, robust_on, true, gt, GS), false, g, red_lds);
            LGPROF(0);
            accumulate_pairs_range(P, 0, P.n_pair, gt, GS);
            grid_sync(g);
            LGPROF(1);
            {   // diagonal blocks / gradients of every vertex from its pairs: eight lanes per entry, each walking every eighth pair of the list (two dependent index
                // loads per pair are pure latency: an object seen from 60 views was 60 round trips in a row), combined by an xor butterfly (csrc/lm_dist.hip: ba_gather_kernel)
                constexpr int SL = 8;
                const int nc = P.n_cam * 27, no = P.n_obj * 27;
                for (int t = gt; t < (nc + no) * SL; t += GS) {
                    const int item = t / SL, sl = t - item * SL;
                    const bool is_cam = item < nc;
                    const int v = is_cam ? item / 27 : (item - nc) / 27, k = is_cam ? item - v * 27 : item - nc - v * 27;
                    const bool fixed = is_cam ? P.cam_fixed[v] != 0 : P.obj_fixed[v] != 0;
                    const int* ptr = is_cam ? P.cam_pair_ptr : P.obj_pair_ptr;
                    const int* lst = is_cam ? P.cam_pair_idx : P.obj_pair_idx;
                    const int off = is_cam ? (k < 21 ? k : 78 + (k - 21)) : (k < 21 ? 21 + k : 84 + (k - 21));
                    double sacc = 0;
                    if (!fixed)
                        for (int j = ptr[v] + sl; j < ptr[v + 1]; j += SL) sacc += P.pair_part[90 * (size_t)lst[j] + off];
#pragma unroll
                    for (int o = 1; o < SL; o <<= 1) sacc += __shfl_xor(sacc, o, 64);
                    if (sl == 0 && !fixed) {
                        if (is_cam) { if (k < 21) P.Hcc[36 * v + k] = sacc; else P.bc[6 * v + (k - 21)] = sacc; }
                        else { if (k < 21) P.Hoo[36 * v + k] = sacc; else P.bo[6 * v + (k - 21)] = sacc; }
                    }
                }
            }
            grid_sync(g);
            LGPROF(2);
            if (it == 0) {      // computeLambdaInit: tau * max |diag|
                double md = 0;
                for (int idx = gt; idx < (P.n_cam + P.n_obj) * 6; idx += GS) {
                    const int v = idx / 6, d = idx - v * 6;
                    if (v < P.n_cam) { if (!P.cam_fixed[v]) md = fmax(md, fabs(P.Hcc[36 * v + diag21[d]])); }
                    else { const int o = v - P.n_cam; if (!P.obj_fixed[o]) md = fmax(md, fabs(P.Hoo[36 * o + diag21[d]])); }
                }
                md = grid_reduce(md, true, g, red_lds);
                lambda = 1e-5 * md;
                ni = 2;
            }
            // ---- trials ----------------------------------------------------------------------
            double rho = 0;
            int qmax = 0;
            bool lam_finite = true;
            do {
                double bad = 0;                                  // any failed factorisation in this workgroup's share
                for (int c = gt; c < P.n_cam; c += GS) P.cam_bak[c] = P.cam[c];          // push()
                for (int o = gt; o < P.n_obj; o += GS) P.obj_bak[o] = P.obj[o];
                for (int c = gt; c < P.n_cam; c += GS) {
                    if (P.cam_fixed[c]) continue;
                    double A[36];
                    unpack_sym21(P.Hcc + 36 * c, A);
                    for (int d = 0; d < 6; ++d) A[d * 7] += lambda;
                    if (schur) {
                        double Ai[36];
                        if (!spd_inverse6(A, Ai)) { bad = 1; for (int i = 0; i < 36; ++i) Ai[i] = 0; }
                        for (int i = 0; i < 36; ++i) P.Hcc_inv[36 * c + i] = Ai[i];
                        for (int r = 0; r < 6; ++r) {
                            double sacc = 0;
                            for (int k = 0; k < 6; ++k) sacc += Ai[r * 6 + k] * P.bc[6 * c + k];
                            P.yc[6 * c + r] = sacc;
                        }
                    } else {
                        double x[6] = {0, 0, 0, 0, 0, 0};
                        if (!spd_solve6(A, P.bc + 6 * c, x)) bad = 1;
                        for (int r = 0; r < 6; ++r) P.yc[6 * c + r] = x[r];
                    }
                }
                grid_sync(g);
                LGPROF(3);
                if (!schur) {
                    for (int o = gt; o < P.n_obj; o += GS) {
                        if (P.obj_fixed[o]) continue;
                        double A[36], x[6] = {0, 0, 0, 0, 0, 0};
                        unpack_sym21(P.Hoo + 36 * o, A);
                        for (int d = 0; d < 6; ++d) A[d * 7] += lambda;
                        if (!spd_solve6(A, P.bo + 6 * o, x)) bad = 1;
                        for (int r = 0; r < 6; ++r) P.xo[6 * o + r] = x[r];
                    }
                    for (int c = gt; c < P.n_cam; c += GS) {                // x_c = y_c, applied by the same thread
                        for (int r = 0; r < 6; ++r) P.xc[6 * c + r] = P.cam_fixed[c] ? 0.0 : P.yc[6 * c + r];
                        if (!P.cam_fixed[c]) pose_oplus(P.cam[c], P.xc + 6 * c);
                    }
                } else {
                    // Y[p] = Hcc^-1 Hco[p]
                    for (int idx = gt; idx < P.n_pair * 36; idx += GS) {
                        const int p = idx / 36, rc = idx - p * 36, r = rc / 6, cc = rc - r * 6;
                        const int c = P.pair_cam[p];
                        double s = 0;
                        if (!P.cam_fixed[c] && !P.obj_fixed[P.pair_obj[p]]) {
                            const double* Hco = P.pair_part + 90 * (size_t)p + 42;
                            for (int k = 0; k < 6; ++k) s += P.Hcc_inv[36 * c + r * 6 + k] * Hco[k * 6 + cc];
                        }
                        P.Y[idx] = s;
                    }
                    grid_sync(g);
                    LGPROF(4);
                    // reduced system in HBM:
                    //   S = blockdiag(Hoo + lambda I) - sum_c Hco(c,o1)^T Y(c,o2);   rhs = b_o - sum_c Hco(c,o)^T y_c
                    // Only the lower triangle is referenced by the factorisation.  Eight adjacent lanes share an entry (or a
                    // row of rhs): each sums every eighth camera of o1 and the partial sums are combined by a fixed shuffle
                    // tree.  The partner pair of (c, o2) comes from the dense cam_obj_pair table, and the index arrays this
                    // loop chases (three dependent L2 round trips per camera otherwise) are read from their LDS copies.
                    const int n_low = ns * (ns + 1) / 2;
                    for (int q8 = gt; q8 < 8 * (n_low + ns); q8 += GS) {        // GS is a multiple of 8: an octet stays together
                        const int t = q8 >> 3, part = q8 & 7;
                        const bool is_rhs = t >= n_low;
                        int row, col = 0;
                        if (is_rhs) {
                            row = t - n_low;
                        } else {                                                  // t = row (row + 1) / 2 + col, col <= row
                            row = (int)((sqrt(8.0 * t + 1.0) - 1.0) * 0.5);
                            while ((row + 1) * (row + 2) / 2 <= t) ++row;
                            while (row * (row + 1) / 2 > t) --row;
                            col = t - row * (row + 1) / 2;
                        }
                        const int s1 = row / 6, i = row - s1 * 6, s2 = col / 6, j = col - s2 * 6;
                        const int o1 = l_slot_obj[s1], o2 = l_slot_obj[s2];
                        double acc = 0;
                        for (int a = P.obj_pair_ptr[o1] + part; a < P.obj_pair_ptr[o1 + 1]; a += 8) {
                            const int p1 = x_obj_pair_idx[a], c = x_pair_cam[p1];
                            if (P.cam_fixed[c]) continue;
                            const double* H1 = P.pair_part + 90 * (size_t)p1 + 42;
                            if (is_rhs) {
                                for (int k = 0; k < 6; ++k) acc += H1[k * 6 + i] * P.yc[6 * c + k];
                            } else {
                                const int p2 = x_cam_obj[c * P.n_obj + o2];
                                if (p2 < 0) continue;
                                const double* Y2 = P.Y + 36 * (size_t)p2;
                                for (int k = 0; k < 6; ++k) acc += H1[k * 6 + i] * Y2[k * 6 + j];
                            }
                        }
                        acc += __shfl_down(acc, 4, 8);
                        acc += __shfl_down(acc, 2, 8);
                        acc += __shfl_down(acc, 1, 8);
                        if (part == 0) {
                            if (is_rhs) {
                                sc.S[t] = P.bo[6 * o1 + i] - acc;                 // packed: [lower triangle by rows | rhs]
                            } else {
                                double diag = 0;
                                if (s1 == s2) {
                                    const int rr = i < j ? i : j, c2 = i < j ? j : i;
                                    diag = P.Hoo[36 * o1 + rr * 6 - rr * (rr - 1) / 2 + (c2 - rr)] + (i == j ? lambda : 0.0);
                                }
                                sc.S[t] = diag - acc;
                            }
                        }
                    }
                    grid_sync(g);
                    LGPROF(5);
                    {   // Cholesky + substitutions by one wave of EVERY workgroup: identical inputs, identical instruction
                        // stream, identical x_o everywhere -- which saves the grid barrier that would publish workgroup 0's
                        const int sp = ns | 1;                           // odd LDS pitch (lm_device.h: wg_cholesky_solve)
                        // the packed system: all of a thread's loads are issued before the first LDS store (one L2 round trip)
                        constexpr int NCP = (LM_NS * (LM_NS + 1) / 2 + LM_NS + LG_THREADS - 1) / LG_THREADS;
                        double v[NCP];
#pragma unroll
                        for (int u = 0; u < NCP; ++u) { const int t = tid + u * LG_THREADS; v[u] = t < n_low + ns ? sc.S[t] : 0.0; }
#pragma unroll
                        for (int u = 0; u < NCP; ++u) {
                            const int t = tid + u * LG_THREADS;
                            if (t < n_low) {
                                int row = (int)((sqrt(8.0 * t + 1.0) - 1.0) * 0.5);
                                while ((row + 1) * (row + 2) / 2 <= t) ++row;
                                while (row * (row + 1) / 2 > t) --row;
                                S[row * sp + t - row * (row + 1) / 2] = v[u];
                            } else if (t < n_low + ns) {
                                rhs[t - n_low] = v[u];
                            }
                        }
                        if (tid == 0) sh_ok = 1;
                        __syncthreads();
                        LGPROF(10);
                        wg_cholesky_solve(S, sp, rhs, ns, tid, LG_THREADS, &sh_ok);
                        __syncthreads();
                        LGPROF(11);
                        // solution (every workgroup writes the same words)
                        if (sh_ok == 0) bad = 1;
                        for (int idx = tid; idx < P.n_obj * 6; idx += LG_THREADS) {
                            const int o = idx / 6;
                            P.xo[idx] = P.obj_slot[o] >= 0 ? rhs[6 * P.obj_slot[o] + (idx - o * 6)] : 0.0;
                        }
                    }
                    __syncthreads();                                     // this workgroup's x_o is visible to its threads
                    LGPROF(6);
                    // x_c = y_c - sum_o Y(c,o) x_o and the camera update: eight adjacent lanes per camera, lane r < 6 computes
                    // row r, lane 0 collects the six rows by shuffles and applies them -- no barrier between the two
                    for (int q8 = gt; q8 < 8 * P.n_cam; q8 += GS) {          // GS is a multiple of 8: an octet stays together
                        const int c = q8 >> 3, r = q8 & 7;
                        const bool free_cam = !P.cam_fixed[c];
                        double s = 0;
                        if (r < 6 && free_cam) {
                            s = P.yc[6 * c + r];
                            for (int b = P.cam_pair_ptr[c]; b < P.cam_pair_ptr[c + 1]; ++b) {
                                const int p = P.cam_pair_idx[b], o = P.pair_obj[p];
                                if (P.obj_fixed[o]) continue;
                                for (int k = 0; k < 6; ++k) s -= P.Y[36 * (size_t)p + r * 6 + k] * P.xo[6 * o + k];
                            }
                        }
                        if (r < 6) P.xc[6 * c + r] = s;
                        double x[6];
                        for (int k = 0; k < 6; ++k) x[k] = __shfl(s, k, 8);
                        if (r == 0 && free_cam) pose_oplus(P.cam[c], x);
                    }
                }
                // The update is applied whether or not a factorisation failed anywhere: a failure is reduced together with
                // chi2 and the step scale below and turns the trial into a rejection, whose pop() restores the poses.  One
                // grid barrier (poses -> edge pass) and one three-value reduction per trial instead of four barriers.
                for (int o = gt; o < P.n_obj; o += GS) if (!P.obj_fixed[o]) pose_oplus(P.obj[o], P.xo + 6 * o);
                grid_sync(g);
                LGPROF(7);
                double red3[LG_RED_N] = {edge_pass_partial(P, 0, P.n_edge, robust_on, false, gt, GS), 0.0, bad};
                for (int idx = gt; idx < P.n_cam * 6; idx += GS)
                    if (!P.cam_fixed[idx / 6]) red3[1] += P.xc[idx] * (lambda * P.xc[idx] + P.bc[idx]);
                for (int idx = gt; idx < P.n_obj * 6; idx += GS)
                    if (!P.obj_fixed[idx / 6]) red3[1] += P.xo[idx] * (lambda * P.xo[idx] + P.bo[idx]);
                grid_reduce3(red3, g, red_lds);
                LGPROF(8);
                const bool ok2 = red3[2] == 0;
                const double tempChi = ok2 ? red3[0] : 1.7976931348623157e308;
                const double scl = ok2 ? red3[1] : 0.0;
                rho = (currentChi - tempChi) / (scl + 1e-3);
                if (rho > 0 && isfinite(tempChi)) {
                    double alpha = 1. - pow(2 * rho - 1, 3.0);
                    alpha = fmin(alpha, 2. / 3.);
                    lambda *= fmax(1. / 3., alpha);
                    ni = 2;
                    currentChi = tempChi;
                } else {
                    lambda *= ni;
                    ni *= 2;
                    for (int c = gt; c < P.n_cam; c += GS) P.cam[c] = P.cam_bak[c];     // pop()
                    for (int o = gt; o < P.n_obj; o += GS) P.obj[o] = P.obj_bak[o];
                    grid_sync(g);
                    if (!isfinite(lambda)) { lam_finite = false; break; }
                }
                ++qmax;
                ++lm_trials;
            } while (rho < 0 && qmax < 10);
            ++lm_its;
            if (qmax == 10 || rho == 0 || !lam_finite) break;
        }
        // ---- re-classification (object_slam.py:877-896), chi2 at the accepted state -----------
        double my_good = 0;
        for (int e = gt; e < P.n_edge; e += GS) {
            double er[2];
            edge_error(P, e, er, nullptr, nullptr);
            const double c2 = edge_chi2(P, e, er);
            P.edge_chi2[e] = c2;
            if (c2 > P.chi2_thr) { P.level[e] = 1; P.edge_inlier[e] = 0; }
            else { P.level[e] = 0; P.edge_inlier[e] = 1; my_good += 1; }
        }
        num_good = (int)grid_reduce(my_good, false, g, red_lds);
        if (round == drop) robust_on = false;
    }
    for (int c = gt; c < P.n_cam; c += GS) pose_to_T(P.cam[c], P.cam_T + 12 * c);
    for (int o = gt; o < P.n_obj; o += GS) pose_to_T(P.obj[o], P.obj_T + 12 * o);
    if (gt == 0) { P.stats[0] = rounds; P.stats[1] = lm_its; P.stats[2] = lm_trials; P.stats[3] = num_good; }
}

size_t lm_grid_scratch_bytes() { return 64 + 2 * LG_RED_N * LG_MAX_WGS * sizeof(double) + (LM_NS * LM_NS + LM_NS) * sizeof(double); }

// `scratch_dev`: lm_grid_scratch_bytes() of device memory whose first 64 bytes are ZERO (the barrier words)
int launch_lm_grid(const void* problem_dev, void* scratch_dev, int n_wgs, hipStream_t s) {
    if (n_wgs < 1) n_wgs = 1;
    if (n_wgs > LG_MAX_WGS) n_wgs = LG_MAX_WGS;
    LmGridScratch sc;
    sc.bar = (unsigned*)scratch_dev;
    sc.red = (double*)((char*)scratch_dev + 64);
    sc.S = sc.red + 2 * LG_RED_N * LG_MAX_WGS;
    hipLaunchKernelGGL(lm_grid_kernel, dim3(8 * n_wgs), dim3(LG_THREADS), 0, s, (const LmProblem*)problem_dev, sc, n_wgs);
    SUO_HIP_CHECK(hipGetLastError());
    return SUO_OK;
}

}  // namespace suo

#ifdef SUO_LG_PROFILE
// profile builds only: microseconds per section since the last call (and reset)
extern "C" int suo_debug_lg_prof(double* out16) {
    long long h[16], z[16] = {0};
    if (hipMemcpyFromSymbol(h, HIP_SYMBOL(suo::g_lg_prof), sizeof(h)) != hipSuccess) return 1;
    if (hipMemcpyToSymbol(HIP_SYMBOL(suo::g_lg_prof), z, sizeof(z)) != hipSuccess) return 1;
    for (int i = 0; i < 16; ++i) out16[i] = h[i] * 0.01;          // s_memrealtime ticks of 10 ns
    return 0;
}
#endif
