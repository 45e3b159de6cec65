// Device-side building blocks of the LM / bundle-adjustment kernels (shared by csrc/lm.hip and csrc/lm_dist.hip).
// See csrc/lm.hip for the reference citations.
#pragma once
#include <type_traits>

#include "suo_internal.h"

namespace suo {

#define DEV __device__ __forceinline__

// threads of the workgroup that owns one problem: 256 (4 waves; frame-sized graphs share a CU with the CNN) or, for the
// large SLAM graphs, 1024 (16 waves, csrc/lm_big.hip compiles the same source with SUO_LM_THREADS=1024)
#ifndef SUO_LM_THREADS
#define SUO_LM_THREADS 256
#endif
constexpr int LM_THREADS = SUO_LM_THREADS;
constexpr int LM_MAX_SCHUR_OBJ = 16;                 // reduced system <= 96 x 96 doubles in LDS
constexpr int LM_NS = 6 * LM_MAX_SCHUR_OBJ;

struct Pose { double q[4]; double t[3]; };

DEV void q_to_R(const double* q, double* R) {
    const double tx = 2 * q[1], ty = 2 * q[2], tz = 2 * q[3];
    const double twx = tx * q[0], twy = ty * q[0], twz = tz * q[0];
    const double txx = tx * q[1], txy = ty * q[1], txz = tz * q[1];
    const double tyy = ty * q[2], tyz = tz * q[2], tzz = tz * q[3];
    R[0] = 1 - (tyy + tzz); R[1] = txy - twz;       R[2] = txz + twy;
    R[3] = txy + twz;       R[4] = 1 - (txx + tzz); R[5] = tyz - twx;
    R[6] = txz - twy;       R[7] = tyz + twx;       R[8] = 1 - (txx + tyy);
}

// 1 / sqrt(s) to the last bit or two: v_rsq_f64 (~27 bits) + two Newton steps + one residual correction, ~14 instructions where
// sqrt() followed by a division is ~60 -- these kernels are single dependent chains of fp64 instructions, so the count IS the time
DEV double rsqrt_nr(double s) {
    double y = __builtin_amdgcn_rsq(s);
    const double h = 0.5 * s;
    y = y * fma(-(h * y), y, 1.5);
    y = y * fma(-(h * y), y, 1.5);
    const double e = fma(-(s * y), y, 1.0);
    return fma(0.5 * y, e, y);
}

DEV void R_to_q(const double* R, double* q) {
    double t = R[0] + R[4] + R[8];
    if (t > 0) {
        const double r = rsqrt_nr(t + 1.0);                  // t = sqrt(t + 1); q0 = t / 2; the others times 1 / (2 t)
        q[0] = 0.5 * ((t + 1.0) * r);
        t = 0.5 * r;
        q[1] = (R[7] - R[5]) * t; q[2] = (R[2] - R[6]) * t; q[3] = (R[3] - R[1]) * t;
    } else {
        int i = 0;
        if (R[4] > R[0]) i = 1;
        if (R[8] > R[i * 4]) i = 2;
        const int j = (i + 1) % 3, k = (j + 1) % 3;
        t = sqrt(R[i * 4] - R[j * 4] - R[k * 4] + 1.0);
        double v[3];
        v[i] = 0.5 * t;
        t = 0.5 / t;
        q[0] = (R[k * 3 + j] - R[j * 3 + k]) * t;
        v[j] = (R[j * 3 + i] + R[i * 3 + j]) * t;
        v[k] = (R[k * 3 + i] + R[i * 3 + k]) * t;
        q[1] = v[0]; q[2] = v[1]; q[3] = v[2];
    }
    if (q[0] < 0) { q[0] = -q[0]; q[1] = -q[1]; q[2] = -q[2]; q[3] = -q[3]; }
    const double inv = rsqrt_nr(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);      // (one reciprocal root instead of a root and four divisions)
    q[0] *= inv; q[1] *= inv; q[2] *= inv; q[3] *= inv;
}

DEV void q_mul(const double* a, const double* b, double* o) {
    o[0] = a[0] * b[0] - a[1] * b[1] - a[2] * b[2] - a[3] * b[3];
    o[1] = a[0] * b[1] + a[1] * b[0] + a[2] * b[3] - a[3] * b[2];
    o[2] = a[0] * b[2] + a[2] * b[0] + a[3] * b[1] - a[1] * b[3];
    o[3] = a[0] * b[3] + a[3] * b[0] + a[1] * b[2] - a[2] * b[1];
}

DEV void pose_from_T(const double* T, Pose& p) {
    const double R[9] = {T[0], T[1], T[2], T[4], T[5], T[6], T[8], T[9], T[10]};
    R_to_q(R, p.q);
    p.t[0] = T[3]; p.t[1] = T[7]; p.t[2] = T[11];
}
DEV void pose_to_T(const Pose& p, double* T) {
    double R[9];
    q_to_R(p.q, R);
    for (int r = 0; r < 3; ++r) { T[4 * r] = R[3 * r]; T[4 * r + 1] = R[3 * r + 1]; T[4 * r + 2] = R[3 * r + 2]; T[4 * r + 3] = p.t[r]; }
}

// T <- exp([omega, upsilon]) * T   (SE3Quat::exp + operator*, thirdparty/g2opy/g2o/types/slam3d/se3quat.h:220-254, :101-114)
// Rodrigues coefficients a = sin(th) / th, b = (1 - cos(th)) / th^2, c = (th - sin(th)) / th^3 of R = I + a Om + b Om^2, V = I + b Om + c Om^2:
//   th < 1e-5          g2o's own shortcut R = V = I + Om + Om^2 (se3quat.h:236-240), kept as it is;
//   th^2 < 0.09        their Taylor series in th^2 (eight terms: < 1e-17 relative) -- no square root, no sin / cos, no divisions; these
//                      are the values the closed forms approximate (the closed forms lose digits of b, c to cancellation at small th);
//   else               the closed forms.
// Om^2 is written out (w w^T - th^2 I with the diagonal as sums of two squares): the same values as the 3x3 product, whose 18 products by
// a literal zero the compiler may not drop.
DEV void pose_oplus(Pose& p, const double* u) {
    const double* w = u;
    const double* ups = u + 3;
    const double w00 = w[0] * w[0], w11 = w[1] * w[1], w22 = w[2] * w[2];
    const double th2 = (w00 + w11) + w22;
    const double Om[9] = {0, -w[2], w[1], w[2], 0, -w[0], -w[1], w[0], 0};
    const double w01 = w[0] * w[1], w02 = w[0] * w[2], w12 = w[1] * w[2];
    const double Om2[9] = {-(w22 + w11), w01, w02, w01, -(w22 + w00), w12, w02, w12, -(w11 + w00)};
    double a, b, b2, c;                                       // R = I + a Om + b Om2,  V = I + b2 Om + c Om2
    if (th2 < 1e-10) {
        a = 1; b = 1; b2 = 1; c = 1;
    } else if (th2 < 0.09) {
        const double t = th2;
        a = fma(t, fma(t, fma(t, fma(t, fma(t, fma(t, fma(t, -1.0 / 1307674368000.0, 1.0 / 6227020800.0), -1.0 / 39916800.0), 1.0 / 362880.0), -1.0 / 5040.0), 1.0 / 120.0), -1.0 / 6.0), 1.0);
        b = fma(t, fma(t, fma(t, fma(t, fma(t, fma(t, fma(t, -1.0 / 20922789888000.0, 1.0 / 87178291200.0), -1.0 / 479001600.0), 1.0 / 3628800.0), -1.0 / 40320.0), 1.0 / 720.0), -1.0 / 24.0), 0.5);
        c = fma(t, fma(t, fma(t, fma(t, fma(t, fma(t, fma(t, -1.0 / 355687428096000.0, 1.0 / 1307674368000.0), -1.0 / 6227020800.0), 1.0 / 39916800.0), -1.0 / 362880.0), 1.0 / 5040.0), -1.0 / 120.0), 1.0 / 6.0);
        b2 = b;
    } else {
        // (theta^3 as two multiplications: the libm pow() of se3quat.h:247 costs ~200 instructions per trial and differs from this by <= 1 ulp)
        const double theta = sqrt(th2), st = sin(theta);
        a = st / theta; b = (1 - cos(theta)) / (theta * theta); c = (theta - st) / (theta * theta * theta);
        b2 = b;
    }
    double R[9], V[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) {
        R[i] = (i % 4 == 0 ? 1.0 : 0.0) + a * Om[i] + b * Om2[i];
        V[i] = (i % 4 == 0 ? 1.0 : 0.0) + b2 * Om[i] + c * Om2[i];
    }
    double eq[4], et[3];
    R_to_q(R, eq);
    for (int r = 0; r < 3; ++r) et[r] = V[3 * r] * ups[0] + V[3 * r + 1] * ups[1] + V[3 * r + 2] * ups[2];
    double Re[9], nt[3], nq[4];
    q_to_R(eq, Re);
    for (int r = 0; r < 3; ++r) nt[r] = et[r] + Re[3 * r] * p.t[0] + Re[3 * r + 1] * p.t[1] + Re[3 * r + 2] * p.t[2];
    q_mul(eq, p.q, nq);
    if (nq[0] < 0) { nq[0] = -nq[0]; nq[1] = -nq[1]; nq[2] = -nq[2]; nq[3] = -nq[3]; }
    const double inv = rsqrt_nr(nq[0] * nq[0] + nq[1] * nq[1] + nq[2] * nq[2] + nq[3] * nq[3]);
    for (int k = 0; k < 4; ++k) p.q[k] = nq[k] * inv;
    for (int k = 0; k < 3; ++k) p.t[k] = nt[k];
}

DEV double huber_rho(double e2, double delta, double& rho1) {
    const double dsqr = delta * delta;
    if (e2 <= dsqr) { rho1 = 1.0; return e2; }
    const double sq = sqrt(e2);
    rho1 = delta / sq;
    return 2 * sq * delta - dsqr;
}

// Wave-wide sum, the same value in every lane.  Inside a row of 16 lanes the partners are reached by DPP register moves (quad
// permutes for lane ^ 1 and lane ^ 2, then the half-row and row mirrors), the four row sums are read with v_readlane and added in
// row order: no LDS crossbar round trips.  (__shfl_xor on a double is two ds_bpermute_b32 per step, ~120 cycles of dependent latency
// each: six steps x 27 sums were 43 % of an LM iteration of the frame kernel, csrc/lm_frame.hip.)
template <int CTRL>
DEV double dpp_get(double v) {
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
DEV double wsum(double v) {
    v += dpp_get<0xB1>(v);            // quad_perm [1,0,3,2]: lane ^ 1
    v += dpp_get<0x4E>(v);            // quad_perm [2,3,0,1]: lane ^ 2
    v += dpp_get<0x141>(v);           // row_half_mirror: the other quad of the half row
    v += dpp_get<0x140>(v);           // row_mirror: the other half row -> every lane of a row holds the row sum
    const int lo = __double2loint(v), hi = __double2hiint(v);
    const double r0 = __hiloint2double(__builtin_amdgcn_readlane(hi, 0), __builtin_amdgcn_readlane(lo, 0));
    const double r1 = __hiloint2double(__builtin_amdgcn_readlane(hi, 16), __builtin_amdgcn_readlane(lo, 16));
    const double r2 = __hiloint2double(__builtin_amdgcn_readlane(hi, 32), __builtin_amdgcn_readlane(lo, 32));
    const double r3 = __hiloint2double(__builtin_amdgcn_readlane(hi, 48), __builtin_amdgcn_readlane(lo, 48));
    return ((r0 + r1) + r2) + r3;
}

// deterministic workgroup sum / max (fixed order); `red` = LM_THREADS/64 doubles of LDS
DEV double block_sum(double v, double* red) {
    v = wsum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    double s = 0;
    for (int i = 0; i < LM_THREADS / 64; ++i) s += red[i];
    return s;
}
DEV double block_max(double v, double* red) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o, 64));
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    double s = red[0];
    for (int i = 1; i < LM_THREADS / 64; ++i) s = fmax(s, red[i]);
    return s;
}

// in-place inverse of a symmetric positive definite 6x6 by Cholesky; false if not PD
__device__ bool spd_inverse6(const double* A, double* Ainv) {
    double L[36];
    for (int i = 0; i < 36; ++i) L[i] = 0;
    for (int i = 0; i < 6; ++i)
        for (int j = 0; j <= i; ++j) {
            double s = A[i * 6 + j];
            for (int k = 0; k < j; ++k) s -= L[i * 6 + k] * L[j * 6 + k];
            if (i == j) { if (!(s > 0) || !isfinite(s)) return false; L[i * 6 + i] = sqrt(s); }
            else L[i * 6 + j] = s / L[j * 6 + j];
        }
    for (int c = 0; c < 6; ++c) {          // solve A x = e_c
        double y[6], x[6];
        for (int i = 0; i < 6; ++i) { double s = (i == c) ? 1.0 : 0.0; for (int k = 0; k < i; ++k) s -= L[i * 6 + k] * y[k]; y[i] = s / L[i * 6 + i]; }
        for (int i = 5; i >= 0; --i) { double s = y[i]; for (int k = i + 1; k < 6; ++k) s -= L[k * 6 + i] * x[k]; x[i] = s / L[i * 6 + i]; }
        for (int i = 0; i < 6; ++i) Ainv[i * 6 + c] = x[i];
    }
    return true;
}

// column `col` of the same inverse (the factorisation repeated, ONE of the six solves): the operations of spd_inverse6 for that column, bit for bit -- for kernels that
// give a lane to each column (csrc/lm_dist.hip: ba_schur_cams_kernel)
__device__ bool spd_inverse6_col(const double* A, int col, double* x) {
    double L[36];
#pragma unroll
    for (int i = 0; i < 36; ++i) L[i] = 0;
    bool ok = true;
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
        for (int j = 0; j <= i; ++j) {
            double s = A[i * 6 + j];
#pragma unroll
            for (int k = 0; k < j; ++k) s -= L[i * 6 + k] * L[j * 6 + k];
            if (i == j) { if (!(s > 0) || !isfinite(s)) ok = false; L[i * 6 + i] = sqrt(ok ? s : 1.0); }
            else L[i * 6 + j] = s / L[j * 6 + j];
        }
    double y[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        double s = (i == col) ? 1.0 : 0.0;
#pragma unroll
        for (int k = 0; k < i; ++k) s -= L[i * 6 + k] * y[k];
        y[i] = s / L[i * 6 + i];
    }
#pragma unroll
    for (int i = 5; i >= 0; --i) {
        double s = y[i];
#pragma unroll
        for (int k = i + 1; k < 6; ++k) s -= L[k * 6 + i] * x[k];
        x[i] = s / L[i * 6 + i];
    }
    return ok;
}

struct LmProblem {
    // sizes
    int n_cam, n_obj, n_edge, n_pair;
    // poses (row-major 3x4), in/out
    double* cam_T; double* obj_T;
    const uint8_t* cam_fixed; const uint8_t* obj_fixed;
    // edges, sorted by pair
    const int* edge_pair;                    // [n_edge]
    const double* edge_k; const double* edge_p; const double* edge_uv; const double* edge_info;
    uint8_t* edge_inlier; double* edge_chi2; // in/out, out
    // pairs
    const int* pair_cam; const int* pair_obj; const int* pair_start;   // pair_start [n_pair+1]
    const int* cam_pair_ptr; const int* cam_pair_idx;                  // CSR camera -> pairs
    const int* obj_pair_ptr; const int* obj_pair_idx;                  // CSR object -> pairs
    const int* cam_obj_pair;                                           // [n_cam][n_obj] pair of (camera, object) or -1
    // parameters
    int its[8]; int n_rounds; int init_with_outliers; double chi2_thr; double huber_delta;
    // scratch (device)
    Pose* cam; Pose* obj; Pose* cam_bak; Pose* obj_bak;
    double* err;                 // [n_edge][2]
    double* jac;                 // [n_edge][29]: Jc(12) Jo(12) rho'*Omega (xx,xy,yy) omega_r(2)
    uint8_t* level;              // [n_edge]
    double* pair_part;           // [n_pair][90]: Hcc(21) Hoo(21) Hco(36) bc(6) bo(6)
    double* Hcc; double* bc;     // [n_cam][36], [n_cam][6]
    double* Hoo; double* bo;     // [n_obj][36], [n_obj][6]
    double* Hcc_inv;             // [n_cam][36]
    double* Y;                   // [n_pair][36]  = Hcc_inv * Hco
    double* yc;                  // [n_cam][6]    = Hcc_inv * bc
    double* xc; double* xo;      // [n_cam][6], [n_obj][6]
    int* obj_slot;               // [n_obj] position in the reduced system (free objects only) or -1
    int* stats;                  // [4] rounds, LM iterations, LM trials, num_good
    // optional (nullptr = pair p's edges end at pair_start[p + 1]): explicit end of each pair's edge range, for problems built ON THE
    // DEVICE in fixed-stride slots (csrc/frame_geom.hip: 41 slots per object, the first n_keypoints of them used); lm_frame_kernel only
    const int* pair_end;
};
DEV int pair_hi(const LmProblem& P, int p) { return P.pair_end ? P.pair_end[p] : P.pair_start[p + 1]; }

// Cholesky factorisation, forward and backward substitution of the ns x ns reduced (object) system by the WHOLE workgroup, blocked
// by 6 (one object's pose block; ns is a multiple of 6): right-looking Cholesky of the lower triangle in LDS (row-major, pitch `sp`
// doubles, only the lower triangle is referenced -- use an ODD pitch: with sp = ns = 48 a column access lands on one LDS bank), then
// blocked forward / backward substitution; rhs is overwritten by the solution.  Per block step every thread factors the 6 x 6 diagonal
// block redundantly in registers (broadcast LDS reads, no hand-off), thread t solves panel row t against it, one barrier, the
// trailing update is tiled 16 x 16 over the threads, one barrier: 2 ns / 6 barriers for the factorisation.  (Until round 2 this was
// a one-wave column loop -- ns dependent steps, each a dot product of LDS round trips: 161 us for a 96-row system, a third of a
// bundle-adjustment trial.)  *ok is cleared on a non-positive / non-finite pivot; the factorisation then continues with a unit pivot
// and the caller rejects the trial.  Call from ALL threads of the workgroup (>= 256); `rhs` holds the solution after the caller's
// next barrier.
DEV void wg_cholesky_solve(double* S, int sp, double* rhs, int ns_, int tid, int nthreads, int* ok) {
    const int ns = __builtin_amdgcn_readfirstlane(ns_);
    auto load_diag = [&](int k0, double (&L)[6][6]) {
#pragma unroll
        for (int r = 0; r < 6; ++r)
#pragma unroll
            for (int c = 0; c <= r; ++c) L[r][c] = S[(k0 + r) * sp + k0 + c];
    };
    for (int k0 = 0; k0 < ns; k0 += 6) {
        double L[6][6], rd[6];
        load_diag(k0, L);
        bool good = true;
#pragma unroll
        for (int c = 0; c < 6; ++c) {
            double piv = L[c][c];
#pragma unroll
            for (int q = 0; q < c; ++q) piv -= L[c][q] * L[c][q];
            if (!(piv > 0) || !isfinite(piv)) good = false;
            const double pp = piv > 0 ? piv : 1.0;
            rd[c] = rsqrt(pp);
            L[c][c] = pp * rd[c];
#pragma unroll
            for (int r = c + 1; r < 6; ++r) {
                double v = L[r][c];
#pragma unroll
                for (int q = 0; q < c; ++q) v -= L[r][q] * L[c][q];
                L[r][c] = v * rd[c];
            }
        }
        if (!good && tid == 0) *ok = 0;
        // panel row i = k0 + 6 + t:  L21[i][.] = A21[i][.] L11^-T
        for (int i = k0 + 6 + tid; i < ns; i += nthreads) {
            double x[6];
#pragma unroll
            for (int c = 0; c < 6; ++c) {
                double v = S[i * sp + k0 + c];
#pragma unroll
                for (int q = 0; q < c; ++q) v -= x[q] * L[c][q];
                x[c] = v * rd[c];
            }
#pragma unroll
            for (int c = 0; c < 6; ++c) S[i * sp + k0 + c] = x[c];
        }
        __syncthreads();                                   // panel complete; every thread has read the diagonal block
        if (tid == 0) {
            // the factored diagonal block back to LDS -- by ONE lane with compile-time indices (27 stores).  Six lanes, each selecting "its" row of L out of
            // registers with a runtime index, were 72 conditional moves in six divergent blocks on wave 0's way into the trailing update (0.4 us per block step).
#pragma unroll
            for (int r = 0; r < 6; ++r) {
#pragma unroll
                for (int c = 0; c <= r; ++c) S[(k0 + r) * sp + k0 + c] = L[r][c];
                // 1 / L_rr for the substitutions, kept in the never-referenced element right of the diagonal (sp > ns: the last row's is
                // the pad column) -- six fp64 divisions per block step and thread otherwise
                S[(k0 + r) * sp + k0 + r + 1] = rd[r];
            }
        }
        // Trailing update of the lower triangle, 16 x 16 tiles of (row a, column b <= a) over the threads.  Per tile row: the (at most six) tiles of the row are
        // loaded together, updated as INDEPENDENT chains and stored together -- the loads are unconditional (clamped rows, results selected) and the six-term
        // chains of a row's tiles interleave; as a loop over tiles with a predicated body every element was its own basic block: LDS round trip -> twelve dependent
        // fp64 operations -> store, one after the other (22 of the 56 us of a 96-row solve, thread 0's wall clock).  Same operations per element: same bits.
        const int m = ns - k0 - 6, base = k0 + 6;
        const int ty = (tid >> 4) & 15, tx = tid & 15;
        if (tid < 256) {
            for (int a0 = 0; a0 < m; a0 += 16) {
                const int a = a0 + ty;
                const bool a_ok = a < m;
                const double* rowa = S + (base + (a_ok ? a : m - 1)) * sp;         // (a row of the system whatever the lane: no predicated loads)
                double la[6];
#pragma unroll
                for (int c = 0; c < 6; ++c) la[c] = rowa[k0 + c];
                double acc[6], lbv[6][6];
                bool ok[6];
#pragma unroll
                for (int j = 0; j < 6; ++j) {
                    const int b = 16 * j + tx;
                    ok[j] = 16 * j <= a0 && a_ok && b <= a;
                    const int bb = b < m ? b : m - 1;
                    acc[j] = rowa[base + bb];
#pragma unroll
                    for (int c = 0; c < 6; ++c) lbv[j][c] = S[(base + bb) * sp + k0 + c];
                }
#pragma unroll
                for (int c = 0; c < 6; ++c)
#pragma unroll
                    for (int j = 0; j < 6; ++j) acc[j] -= la[c] * lbv[j][c];
#pragma unroll
                for (int j = 0; j < 6; ++j)
                    if (ok[j]) S[(base + a) * sp + base + 16 * j + tx] = acc[j];
            }
        }
        __syncthreads();
    }
    // The substitutions by ONE wave (round 5): 2 ns / 6 block steps whose only parallel work is ns rows of six multiply-adds -- with the whole workgroup each step paid a
    // workgroup barrier and four redundant copies of the 6 x 6 solve (16 of the 56 us of a 96-row solve); a wave's LDS operations execute in order, so it needs none.
    // L y = rhs, block by block: every lane solves the 6 x 6 block redundantly, lane t then updates rows t, t + 64 below it
    if (tid < 64) {
        for (int k0 = 0; k0 < ns; k0 += 6) {
            double L[6][6], y[6];
            load_diag(k0, L);
#pragma unroll
            for (int c = 0; c < 6; ++c) {
                double v = rhs[k0 + c];
#pragma unroll
                for (int q = 0; q < c; ++q) v -= L[c][q] * y[q];
                y[c] = v * S[(k0 + c) * sp + k0 + c + 1];
            }
            for (int i = k0 + 6 + tid; i < ns; i += 64) {
                double v = rhs[i];
#pragma unroll
                for (int c = 0; c < 6; ++c) v -= S[i * sp + k0 + c] * y[c];
                rhs[i] = v;
            }
            __builtin_amdgcn_wave_barrier();               // (every lane has read rhs[k0 ..] before lane 0 overwrites it)
            if (tid == 0) {
#pragma unroll
                for (int c = 0; c < 6; ++c) rhs[k0 + c] = y[c];
            }
            __builtin_amdgcn_wave_barrier();
        }
        // L^T x = y, from the last block up
        for (int k0 = ns - 6; k0 >= 0; k0 -= 6) {
            double L[6][6], x[6];
            load_diag(k0, L);
#pragma unroll
            for (int c = 5; c >= 0; --c) {
                double v = rhs[k0 + c];
#pragma unroll
                for (int q = c + 1; q < 6; ++q) v -= L[q][c] * x[q];
                x[c] = v * S[(k0 + c) * sp + k0 + c + 1];
            }
            for (int i = tid; i < k0; i += 64) {
                double v = rhs[i];
#pragma unroll
                for (int c = 0; c < 6; ++c) v -= S[(k0 + c) * sp + i] * x[c];
                rhs[i] = v;
            }
            __builtin_amdgcn_wave_barrier();
            if (tid == 0) {
#pragma unroll
                for (int c = 0; c < 6; ++c) rhs[k0 + c] = x[c];
            }
            __builtin_amdgcn_wave_barrier();
        }
    }
}

// The same solve for a kernel that does nothing else (csrc/lm_dist.hip: ba_solve_kernel -- the global adjustment's reduced system, up to 96 rows): Cholesky
// factorisation, forward and backward substitution by the WHOLE workgroup (NTHREADS threads, all of them calling), blocked by 6 (one object's pose block; ns is a multiple of 6): right-looking Cholesky of the lower triangle in LDS (row-major, pitch `sp` doubles, only
// the lower triangle is referenced -- use an ODD pitch: with sp = ns = 48 a column access lands on one LDS bank).  S must hold ns + 1 rows: the right-hand side rides
// as row ns of the system, so the forward substitution happens inside the factorisation (its panel row is y's block, its trailing update the elimination) and only
// L^T x = y is left for afterwards.
//
// The trailing matrix lives in REGISTERS (round 6): the (ns + 1) x ns lower triangle is cut into 16 x 16 tiles on a fixed grid, dealt round-robin
// to the waves once (27 tiles at ns = 96: seven per wave of a 256-thread workgroup, 56 registers), each the accumulator of v_mfma_f64_16x16x4_f64 and kept NEGATED,
// so a block step's update  C -= L21 L21^T  is  -C += A B  with A = B = the panel's fragments straight from LDS: two MFMAs (K = 6 as 4 + 2), four LDS reads and no
// write per tile and step.  Per block step: every thread factors the 6 x 6 diagonal block redundantly in registers (broadcast LDS reads, no hand-off), thread t
// solves panel row t against it, one barrier, the tiles that still have a live element are updated, the six columns the NEXT step factors are written back to
// LDS, one barrier.  History of this update, 96 rows, thread 0's clock: one element per thread on the vector pipe with 13 LDS reads each (rounds 2-5) 43 k cycles;
// MFMA tiles read from and written to LDS each step 52 k -- one wave per SIMD pays ~5 cycles per instruction, and the index arithmetic of a tile walk plus 12 LDS
// accesses per tile cost what the vector form cost; tiles in registers: see profiles/r06_cholesky.txt.  fp64 MFMA on gfx950 has the vector pipe's rate (64 cycles per
// 16 x 16 x 4): what it buys here is operand reuse, not FLOPs.
// *ok is cleared on a non-positive / non-finite pivot; the factorisation then continues with a unit pivot and the caller rejects the trial.  `rhs` holds the
// solution after the caller's next barrier.
typedef double lm_f64x4 __attribute__((ext_vector_type(4)));
#ifdef SUO_CHOL_PROF
#define CHOL_T(i) do { if (pt) { const long long now_ = clock64(); pt[i] += now_ - tprev_; tprev_ = now_; } } while (0)
#define CHOL_PT , long long* pt
#define CHOL_NOPT , nullptr
#else
#define CHOL_T(i) do { } while (0)
#define CHOL_PT
#define CHOL_NOPT
#endif
template <int NTHREADS>
DEV void wg_cholesky_solve_mfma(double* S, int sp, double* rhs, int ns_, int tid, int* ok CHOL_PT) {
#ifdef SUO_CHOL_PROF
    long long tprev_ = clock64();
#endif
    constexpr int NW = NTHREADS / 64;
    constexpr int MAXTC = (LM_NS + 15) / 16, MAXT = MAXTC * (MAXTC + 1) / 2 + MAXTC;      // tile rows 0 .. LM_NS / 16 (the last holds the right-hand side)
    constexpr int NTW = (MAXT + NW - 1) / NW;
    const int ns = __builtin_amdgcn_readfirstlane(ns_);
    const int lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, lk = lane >> 4;
    auto load_diag = [&](int k0, double (&L)[6][6]) {
#pragma unroll
        for (int r = 0; r < 6; ++r)
#pragma unroll
            for (int c = 0; c <= r; ++c) L[r][c] = S[(k0 + r) * sp + k0 + c];
    };
    for (int i = tid; i < ns; i += NTHREADS) S[ns * sp + i] = rhs[i];
    __syncthreads();
    // this wave's tiles: t = wv + NW u -> (tile row ta, tile column tb <= ta); lane l of a tile holds -C[16 ta + (l >> 4) + 4 q][16 tb + (l & 15)], q < 4, and
    // supplies A[l & 15][l >> 4] = L21[16 ta + (l & 15)][k + (l >> 4)], B likewise from tile column tb's rows.  Rows past ns / columns past ns - 1 are clamped
    // on the way in (their products land in elements that are never written back).
    const int ntc = (ns + 15) >> 4, ntri = ntc * (ntc + 1) / 2, ntiles = ntri + ((ns & 15) == 0 ? ntc : 0);
    int tta[NTW], ttb[NTW], offA[NTW], offB[NTW], wb[NTW][4];
    lm_f64x4 c[NTW];
#pragma unroll
    for (int u = 0; u < NTW; ++u) {
        int t = wv + NW * u, r = 0;
        if (t >= ntiles) {                                                     // (never live)
            tta[u] = ttb[u] = -64; offA[u] = offB[u] = 0; c[u] = lm_f64x4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int q = 0; q < 4; ++q) wb[u][q] = 2;
            continue;
        }
        if (t >= ntri) { r = ntc; t -= ntri; }
        else { while ((r + 1) * (r + 2) / 2 <= t) ++r; t -= r * (r + 1) / 2; }
        tta[u] = r; ttb[u] = t;
        offA[u] = min(16 * r + li, ns) * sp + lk;
        offB[u] = min(16 * t + li, ns - 1) * sp + lk;
        const int col = min(16 * t + li, ns - 1);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int row = 16 * r + lk + 4 * q, cc = 16 * t + li;
            c[u][q] = -S[min(row, ns) * sp + col];
            wb[u][q] = (row >= cc && row <= ns && cc < ns) ? row * sp + cc : 2;       // (element (0, 2): above the diagonal and not a 1 / L_rr slot -- never read)
        }
    }
    CHOL_T(0);
    for (int k0 = 0; k0 < ns; k0 += 6) {
        double L[6][6], rd[6];
        load_diag(k0, L);
        bool good = true;
#pragma unroll
        for (int c = 0; c < 6; ++c) {
            double piv = L[c][c];
#pragma unroll
            for (int q = 0; q < c; ++q) piv -= L[c][q] * L[c][q];
            if (!(piv > 0) || !isfinite(piv)) good = false;
            const double pp = piv > 0 ? piv : 1.0;
            rd[c] = rsqrt(pp);
            L[c][c] = pp * rd[c];
#pragma unroll
            for (int r = c + 1; r < 6; ++r) {
                double v = L[r][c];
#pragma unroll
                for (int q = 0; q < c; ++q) v -= L[r][q] * L[c][q];
                L[r][c] = v * rd[c];
            }
        }
        if (!good && tid == 0) *ok = 0;
        CHOL_T(1);
        // panel row i = k0 + 6 + t (row ns: the right-hand side):  L21[i][.] = A21[i][.] L11^-T
        for (int i = k0 + 6 + tid; i <= ns; i += NTHREADS) {
            double x[6];
#pragma unroll
            for (int c = 0; c < 6; ++c) {
                double v = S[i * sp + k0 + c];
#pragma unroll
                for (int q = 0; q < c; ++q) v -= x[q] * L[c][q];
                x[c] = v * rd[c];
            }
#pragma unroll
            for (int c = 0; c < 6; ++c) S[i * sp + k0 + c] = x[c];
        }
        CHOL_T(2);
        __syncthreads();                                   // panel complete; every thread has read the diagonal block
        CHOL_T(3);
        if (lane == 0 && wv < 4) {
            // the factored diagonal block (and 1 / L_rr for the back substitution, kept in the never-referenced element right of the diagonal -- sp > ns: the last
            // row's is the pad column -- six fp64 divisions per block step otherwise) back to LDS: lane 0 of waves 0 .. 3, seven stores each with compile-time
            // register indices.  (Six lanes each selecting "its" row with a runtime index: 72 conditional moves in six divergent blocks; one lane alone: 33 stores on
            // the way of its wave into the update below, ~450 cycles that the other waves then waited for at the barrier.)
#define LM_PUT_ROW(r) do { _Pragma("unroll") for (int c_ = 0; c_ <= (r); ++c_) S[(k0 + (r)) * sp + k0 + c_] = L[(r)][c_]; S[(k0 + (r)) * sp + k0 + (r) + 1] = rd[(r)]; } while (0)
            if (wv == 0) { LM_PUT_ROW(5); }
            else if (wv == 1) { LM_PUT_ROW(4); }
            else if (wv == 2) { LM_PUT_ROW(3); LM_PUT_ROW(0); }
            else { LM_PUT_ROW(2); LM_PUT_ROW(1); }
#undef LM_PUT_ROW
        }
        const int base = k0 + 6;
        if (base < ns) {
            // every live tile's fragments first (one LDS round trip for the wave), then the first MFMA of every tile, then the second: no MFMA waits for the one before it
            double a0[NTW], b0[NTW], a1[NTW], b1[NTW];
#pragma unroll
            for (int u = 0; u < NTW; ++u) {
                if (16 * ttb[u] + 15 < base) continue;                        // (scalar) every element of the tile is final -- or the tile does not exist
                const double* pa = S + offA[u] + k0;
                const double* pb = S + offB[u] + k0;
                a0[u] = pa[0]; b0[u] = pb[0]; a1[u] = pa[4]; b1[u] = pb[4];   // (lanes lk >= 2 read two columns of the trailing block there: both zeroed)
            }
#ifdef SUO_CHOL_PROF
            __builtin_amdgcn_s_waitcnt(0); __builtin_amdgcn_sched_barrier(0); CHOL_T(8); __builtin_amdgcn_sched_barrier(0);
#endif
#pragma unroll
            for (int u = 0; u < NTW; ++u)
                if (16 * ttb[u] + 15 >= base) c[u] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0[u], b0[u], c[u], 0, 0, 0);
#pragma unroll
            for (int u = 0; u < NTW; ++u)
                if (16 * ttb[u] + 15 >= base) c[u] = __builtin_amdgcn_mfma_f64_16x16x4f64(lk < 2 ? a1[u] : 0.0, lk < 2 ? b1[u] : 0.0, c[u], 0, 0, 0);
#ifdef SUO_CHOL_PROF
            { double sink = 0; _Pragma("unroll") for (int u = 0; u < NTW; ++u) sink += c[u][0]; asm volatile("" :: "v"(sink)); __builtin_amdgcn_sched_barrier(0); }
#endif
            CHOL_T(4);
            // the columns of the next block step, base .. base + 5, from the registers back to LDS (rows >= column, <= ns): its diagonal block and panel.  The LDS
            // address of every element a lane holds is fixed (wb[] below; elements above the diagonal or below row ns point at element (0, 2), which nothing reads), so a tile costs one lane test and four stores
#pragma unroll
            for (int u = 0; u < NTW; ++u) {
                if (16 * ttb[u] + 15 < base || 16 * ttb[u] > base + 5) continue;      // (scalar)
                if ((unsigned)(16 * ttb[u] + li - base) < 6u) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) S[wb[u][q]] = -c[u][q];
                }
            }
        }
        CHOL_T(5);
        __syncthreads();
        CHOL_T(6);
    }
    // L^T x = y by ONE wave (round 5), from the last block up: 2 ns / 6 block steps whose only parallel work is ns rows of six multiply-adds -- with the whole
    // workgroup each step paid a workgroup barrier and four redundant copies of the 6 x 6 solve; a wave's LDS operations execute in order, so it needs none.
    // Every lane solves the 6 x 6 block redundantly, lane t then updates entries t, t + 64 above it.
    if (tid < 64) {
        for (int i = tid; i < ns; i += 64) rhs[i] = S[ns * sp + i];
        __builtin_amdgcn_wave_barrier();
        for (int k0 = ns - 6; k0 >= 0; k0 -= 6) {
            double L[6][6], x[6];
            load_diag(k0, L);
#pragma unroll
            for (int c = 5; c >= 0; --c) {
                double v = rhs[k0 + c];
#pragma unroll
                for (int q = c + 1; q < 6; ++q) v -= L[q][c] * x[q];
                x[c] = v * S[(k0 + c) * sp + k0 + c + 1];
            }
            for (int i = tid; i < k0; i += 64) {
                double v = rhs[i];
#pragma unroll
                for (int c = 0; c < 6; ++c) v -= S[(k0 + c) * sp + i] * x[c];
                rhs[i] = v;
            }
            __builtin_amdgcn_wave_barrier();
            if (tid == 0) {
#pragma unroll
                for (int c = 0; c < 6; ++c) rhs[k0 + c] = x[c];
            }
            __builtin_amdgcn_wave_barrier();
        }
    }
    CHOL_T(7);
}

DEV bool edge_active(const LmProblem& P, int e) {
    const int p = P.edge_pair[e];
    return P.level[e] == 0 && !(P.cam_fixed[P.pair_cam[p]] && P.obj_fixed[P.pair_obj[p]]);
}

DEV void edge_error(const LmProblem& P, int e, double* err, double* pw_out, double* pc_out) {
    const int p = P.edge_pair[e];
    const Pose& cam = P.cam[P.pair_cam[p]];
    const Pose& obj = P.obj[P.pair_obj[p]];
    double Ro[9], Rc[9], pw[3], pc[3];
    q_to_R(obj.q, Ro);
    q_to_R(cam.q, Rc);
    const double* x = P.edge_p + 3 * e;
#pragma unroll
    for (int r = 0; r < 3; ++r) pw[r] = Ro[3 * r] * x[0] + Ro[3 * r + 1] * x[1] + Ro[3 * r + 2] * x[2] + obj.t[r];
#pragma unroll
    for (int r = 0; r < 3; ++r) pc[r] = Rc[3 * r] * pw[0] + Rc[3 * r + 1] * pw[1] + Rc[3 * r + 2] * pw[2] + cam.t[r];
    const double* k = P.edge_k + 4 * e;
    err[0] = P.edge_uv[2 * e] - (k[0] * pc[0] / pc[2] + k[2]);
    err[1] = P.edge_uv[2 * e + 1] - (k[1] * pc[1] / pc[2] + k[3]);
    if (pw_out) { for (int r = 0; r < 3; ++r) { pw_out[r] = pw[r]; pc_out[r] = pc[r]; } }
}
DEV double edge_chi2(const LmProblem& P, int e, const double* err) {
    const double* I = P.edge_info + 3 * e;
    return err[0] * (I[0] * err[0] + I[1] * err[1]) + err[1] * (I[1] * err[0] + I[2] * err[1]);
}

// computeActiveErrors + activeRobustChi2; with_jac also stores the edge Jacobians (linearizeOplus) and the
// Huber-weighted information / gradient factors used by constructQuadraticForm.  Threads over edges.
// this thread's share of the edges [e_begin, e_end): returns its partial (robustified) chi2
// (t0, stride): this thread's position in the loop -- one workgroup by default, a whole grid in csrc/lm_grid.hip
DEV double edge_pass_partial(const LmProblem& P, int e_begin, int e_end, bool robust_on, bool with_jac, int t0 = threadIdx.x,
                             int stride = LM_THREADS) {
    double c = 0;
    for (int e = e_begin + t0; e < e_end; e += stride) {
        if (!edge_active(P, e)) continue;
        double er[2], pw[3], pc[3];
        edge_error(P, e, er, pw, pc);
        P.err[2 * e] = er[0];
        P.err[2 * e + 1] = er[1];
        const double c2 = edge_chi2(P, e, er);
        double w = 1.0;
        c += robust_on ? huber_rho(c2, P.huber_delta, w) : c2;
        if (with_jac) {
            double Rc[9];
            q_to_R(P.cam[P.pair_cam[P.edge_pair[e]]].q, Rc);
            const double* k = P.edge_k + 4 * e;
            const double PJ[6] = {-(k[0] / pc[2]), 0, k[0] * pc[0] / (pc[2] * pc[2]), 0, -(k[1] / pc[2]), k[1] * pc[1] / (pc[2] * pc[2])};
            double PR[6];
            // (every small loop here is unrolled explicitly: left rolled, Dw / Dc were indexed at run time and lived in scratch memory)
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int cc = 0; cc < 3; ++cc) PR[3 * r + cc] = PJ[3 * r] * Rc[cc] + PJ[3 * r + 1] * Rc[3 + cc] + PJ[3 * r + 2] * Rc[6 + cc];
            const double Dw[18] = {0, pw[2], -pw[1], 1, 0, 0, -pw[2], 0, pw[0], 0, 1, 0, pw[1], -pw[0], 0, 0, 0, 1};
            const double Dc[18] = {0, pc[2], -pc[1], 1, 0, 0, -pc[2], 0, pc[0], 0, 1, 0, pc[1], -pc[0], 0, 0, 0, 1};
            double* J = P.jac + 29 * (size_t)e;
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int cc = 0; cc < 6; ++cc) {
                    J[6 * r + cc] = PJ[3 * r] * Dc[cc] + PJ[3 * r + 1] * Dc[6 + cc] + PJ[3 * r + 2] * Dc[12 + cc];            // Jc
                    J[12 + 6 * r + cc] = PR[3 * r] * Dw[cc] + PR[3 * r + 1] * Dw[6 + cc] + PR[3 * r + 2] * Dw[12 + cc];       // Jo
                }
            const double* I = P.edge_info + 3 * e;
            J[24] = w * I[0]; J[25] = w * I[1]; J[26] = w * I[2];
            J[27] = -(I[0] * er[0] + I[1] * er[1]) * w;
            J[28] = -(I[1] * er[0] + I[2] * er[1]) * w;
        }
    }
    return c;
}
DEV double active_errors_and_chi2(const LmProblem& P, bool robust_on, bool with_jac, double* red) {
    return block_sum(edge_pass_partial(P, 0, P.n_edge, robust_on, with_jac), red);
}

// One thread per (pair, entry): entry k of [Hcc(21) | Hoo(21) | Hco(36) | bc(6) | bo(6)] summed over the pair's
// active edges in edge order (deterministic, no cross-lane reduction).  Pairs [p_begin, p_end).
DEV void accumulate_pairs_range(const LmProblem& P, int p_begin, int p_end, int t0 = threadIdx.x, int stride = LM_THREADS) {
    for (int idx = p_begin * 90 + t0; idx < p_end * 90; idx += stride) {
        const int p = idx / 90, k = idx - p * 90;
        const bool cfree = !P.cam_fixed[P.pair_cam[p]], ofree = !P.obj_fixed[P.pair_obj[p]];
        int a_off, b_off, r, c, kind;      // kind 0: A^T O B block entry (r,c); 1: gradient entry r
        if (k < 21) { if (!cfree) continue; kind = 0; a_off = 0; b_off = 0; int u = k; r = 0; while (u >= 6 - r) { u -= 6 - r; ++r; } c = r + u; }
        else if (k < 42) { if (!ofree) continue; kind = 0; a_off = 12; b_off = 12; int u = k - 21; r = 0; while (u >= 6 - r) { u -= 6 - r; ++r; } c = r + u; }
        else if (k < 78) { if (!(cfree && ofree)) continue; kind = 0; a_off = 0; b_off = 12; r = (k - 42) / 6; c = (k - 42) - r * 6; }
        else if (k < 84) { if (!cfree) continue; kind = 1; a_off = 0; b_off = 0; r = k - 78; c = 0; }
        else { if (!ofree) continue; kind = 1; a_off = 12; b_off = 0; r = k - 84; c = 0; }
        // Every edge's terms are loaded and evaluated unconditionally and SELECTED by the edge's level afterwards (an inactive edge
        // adds 0.0, which leaves the sum bit-identical): with a `continue` on the level every iteration was two dependent L2 round
        // trips (level, then the Jacobian row) that the compiler could not overlap across edges -- this loop was 18-24 % of a global
        // adjustment.
        double s = 0;
        const int e0 = P.pair_start[p], e1 = P.pair_start[p + 1];
#pragma unroll 4
        for (int e = e0; e < e1; ++e) {
            const double* J = P.jac + 29 * (size_t)e;
            const double a0 = J[a_off + r], a1 = J[a_off + 6 + r];
            const double w0 = kind == 0 ? J[24] : 0.0, w1 = kind == 0 ? J[25] : 0.0, w2 = kind == 0 ? J[26] : 0.0;
            const double b0 = kind == 0 ? J[b_off + c] : J[27], b1 = kind == 0 ? J[b_off + 6 + c] : J[28];
            const double term = kind == 0 ? (a0 * w0 + a1 * w1) * b0 + (a0 * w1 + a1 * w2) * b1 : a0 * b0 + a1 * b1;
            s += P.level[e] == 0 ? term : 0.0;
        }
        P.pair_part[idx] = s;
    }
}
DEV void accumulate_pairs(const LmProblem& P) { accumulate_pairs_range(P, 0, P.n_pair); }

// solve the symmetric positive definite 6x6 system A x = b by Cholesky; false if not PD (x is then unspecified).
// Fully unrolled with compile-time indices and inlined: as a plain function with runtime loop indices the factor lived in SCRATCH
// memory (368 bytes per lane in lm_frame_kernel) and every one of its ~100 accesses per call was a memory round trip -- the single
// largest cost of a frame's LM trial.  One reciprocal ROOT per pivot (rsqrt_nr) instead of a root and a division per element (an ulp or two from the divided form).
DEV bool spd_solve6(const double* A, const double* b, double* x) {
    double L[6][6], dinv[6];
    bool ok = true;
#pragma unroll
    for (int i = 0; i < 6; ++i) {
#pragma unroll
        for (int j = 0; j <= i; ++j) {
            double s = A[i * 6 + j];
#pragma unroll
            for (int k = 0; k < j; ++k) s -= L[i][k] * L[j][k];
            if (i == j) {
                if (!(s > 0) || !isfinite(s)) ok = false;
                const double sp = ok ? s : 1.0;
                dinv[i] = rsqrt_nr(sp);                          // 1 / L_ii and L_ii = s / sqrt(s) from one reciprocal root
                L[i][i] = sp * dinv[i];
            } else {
                L[i][j] = s * dinv[j];
            }
        }
    }
    double y[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        double s = b[i];
#pragma unroll
        for (int k = 0; k < i; ++k) s -= L[i][k] * y[k];
        y[i] = s * dinv[i];
    }
#pragma unroll
    for (int i = 5; i >= 0; --i) {
        double s = y[i];
#pragma unroll
        for (int k = i + 1; k < 6; ++k) s -= L[k][i] * x[k];
        x[i] = s * dinv[i];
    }
    return ok;
}

DEV void unpack_sym21(const double* s, double* A) {
    int u = 0;
    for (int r = 0; r < 6; ++r)
        for (int c = r; c < 6; ++c) { A[r * 6 + c] = s[u]; A[c * 6 + r] = s[u]; ++u; }
}

}  // namespace suo
