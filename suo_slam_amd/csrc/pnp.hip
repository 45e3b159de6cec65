// Batched P3P-RANSAC PnP on gfx950: one wavefront per object, one RANSAC hypothesis per lane.
//
// Replaces lambdatwist.pnp (thirdparty/lambdatwist/pnp_python_binding.cpp:32-62), i.e.
//   PNP::compute / evaluate_inlier_set / get4RandomInRange0   pnp_ransac.cpp:188-232, 41-87, 161-183
//   PnpParams::get_iterations                                  parameters.h:76-102
//   p4p                                                        p4p.cpp:11-60
//   p3p_lambdatwist<double,5> + cubick + eigwithknown0 + gauss_newton_refineL
//                                                              lambdatwist/*.h
//   PNP::refine (Ceres LM over unit-quaternion (+) translation) pnp_ransac.cpp:240-326
// as called per object from lib/object_slam.py:25-41,1144.
//
// MI355X-first structure: the reference runs 100-1000 sequential hypotheses per object on one CPU
// thread, one object after the other.  Here every object of the frame is one wave; the wave
// evaluates 64 hypotheses at a time (each lane: sample 4 points with a counter-based generator,
// Lambda-Twist P3P, 4th-point disambiguation, exact inlier count over the object's <=41 points),
// then replays the reference's sequential accept rule (strictly-more-inliers, adaptive iteration
// count) over the 64 results with wave broadcasts, so the outcome is identical to the sequential
// loop.  Refinement is a wave-parallel Levenberg-Marquardt (lanes over points, butterfly reduction
// of the 6x6 normal equations, redundant per-lane Cholesky).  All arithmetic is fp64 with
// contraction off so thresholds fall the same way as in the CPU restatement.
#include "suo_internal.h"
#include "tune.h"

namespace suo {

#define DEV __device__ __forceinline__

DEV double dot3(const double* a, const double* b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
DEV void cross3(const double* a, const double* b, double* c) {
    c[0] = a[1] * b[2] - a[2] * b[1];
    c[1] = a[2] * b[0] - a[0] * b[2];
    c[2] = a[0] * b[1] - a[1] * b[0];
}
DEV void normalize3(double* a) {
    const double si = 1.0 / sqrt(a[0] * a[0] + a[1] * a[1] + a[2] * a[2]);
    a[0] *= si; a[1] *= si; a[2] *= si;
}

DEV bool root2real(double b, double c, double& r1, double& r2) {
    const double v = b * b - 4.0 * c;
    if (v < 0) { r1 = r2 = 0.5 * b; return false; }
    const double y = sqrt(v);
    if (b < 0) { r1 = 0.5 * (-b + y); r2 = 0.5 * (-b - y); }
    else { r1 = 2.0 * c / (-b + y); r2 = 2.0 * c / (-b - y); }
    return true;
}

DEV double cubick(double b, double c, double d) {
    double r0;
    if (b * b >= 3.0 * c) {
        const double v = sqrt(b * b - 3.0 * c);
        const double t1 = (-b - v) / 3.0;
        double k = ((t1 + b) * t1 + c) * t1 + d;
        if (k > 0.0) {
            r0 = t1 - sqrt(-k / (3.0 * t1 + b));
        } else {
            const double t2 = (-b + v) / 3.0;
            k = ((t2 + b) * t2 + c) * t2 + d;
            r0 = t2 + sqrt(-k / (3.0 * t2 + b));
        }
    } else {
        r0 = -b / 3.0;
        if (fabs((3.0 * r0 + 2.0 * b) * r0 + c) < 1e-4) r0 += 1;
    }
    for (unsigned cnt = 0; cnt < 50; ++cnt) {
        const double fx = ((r0 + b) * r0 + c) * r0 + d;
        if (cnt < 7 || fabs(fx) > 1e-13) {
            const double fpx = (3.0 * r0 + 2.0 * b) * r0 + c;
            r0 -= fx / fpx;
        } else break;
    }
    return r0;
}

DEV void eigwithknown0(const double* x, double* E, double* L) {
    L[2] = 0;
    double v3[3] = {x[3] * x[7] - x[6] * x[4], x[6] * x[1] - x[7] * x[0], x[4] * x[0] - x[3] * x[1]};
    normalize3(v3);
    const double x01_squared = x[1] * x[1];
    const double b = -x[0] - x[4] - x[8];
    const double c = -x01_squared - x[2] * x[2] - x[5] * x[5] + x[0] * (x[4] + x[8]) + x[4] * x[8];
    double e1, e2;
    root2real(b, c, e1, e2);
    if (fabs(e1) < fabs(e2)) { const double t = e1; e1 = e2; e2 = t; }
    L[0] = e1; L[1] = e2;
    const double mx0011 = -x[0] * x[4];
    const double prec_0 = x[1] * x[5] - x[2] * x[4];
    const double prec_1 = x[1] * x[2] - x[0] * x[5];
    const double e = e1;
    const double tmp = 1.0 / (e * (x[0] + x[4]) + mx0011 - e * e + x01_squared);
    double a1 = -(e * x[2] + prec_0) * tmp;
    double a2 = -(e * x[5] + prec_1) * tmp;
    const double rnorm = 1.0 / sqrt(a1 * a1 + a2 * a2 + 1.0);
    a1 *= rnorm; a2 *= rnorm;
    const double tmp2 = 1.0 / (e2 * (x[0] + x[4]) + mx0011 - e2 * e2 + x01_squared);
    double a21 = -(e2 * x[2] + prec_0) * tmp2;
    double a22 = -(e2 * x[5] + prec_1) * tmp2;
    const double rnorm2 = 1.0 / sqrt(a21 * a21 + a22 * a22 + 1.0);
    a21 *= rnorm2; a22 *= rnorm2;
    E[0] = a1;    E[1] = a21;    E[2] = v3[0];
    E[3] = a2;    E[4] = a22;    E[5] = v3[1];
    E[6] = rnorm; E[7] = rnorm2; E[8] = v3[2];
}

DEV void gauss_newton_refineL(double* L, double a12, double a13, double a23, double b12, double b13, double b23) {
    for (int i = 0; i < 5; ++i) {
        const double l1 = L[0], l2 = L[1], l3 = L[2];
        const double r1 = l1 * l1 + l2 * l2 + b12 * l1 * l2 - a12;
        const double r2 = l1 * l1 + l3 * l3 + b13 * l1 * l3 - a13;
        const double r3 = l2 * l2 + l3 * l3 + b23 * l2 * l3 - a23;
        if (fabs(r1) + fabs(r2) + fabs(r3) < 1e-10) break;
        const double v0 = 2.0 * l1 + b12 * l2, v1 = 2.0 * l2 + b12 * l1;
        const double v3 = 2.0 * l1 + b13 * l3, v5 = 2.0 * l3 + b13 * l1;
        const double v7 = 2.0 * l2 + b23 * l3, v8 = 2.0 * l3 + b23 * l2;
        const double det = 1.0 / (-v0 * v5 * v7 - v1 * v3 * v8);
        const double J0 = -v5 * v7, J1 = -v1 * v8, J2 = v1 * v5;
        const double J3 = -v3 * v8, J4 = v0 * v8, J5 = -v0 * v5;
        const double J6 = v3 * v7, J7 = -v0 * v7, J8 = -v1 * v3;
        const double n1 = l1 - det * (J0 * r1 + J1 * r2 + J2 * r3);
        const double n2 = l2 - det * (J3 * r1 + J4 * r2 + J5 * r3);
        const double n3 = l3 - det * (J6 * r1 + J7 * r2 + J8 * r3);
        const double r11 = n1 * n1 + n2 * n2 + b12 * n1 * n2 - a12;
        const double r12 = n1 * n1 + n3 * n3 + b13 * n1 * n3 - a13;
        const double r13 = n2 * n2 + n3 * n3 + b23 * n2 * n3 - a23;
        if (fabs(r11) + fabs(r12) + fabs(r13) > fabs(r1) + fabs(r2) + fabs(r3)) break;
        L[0] = n1; L[1] = n2; L[2] = n3;
    }
}

DEV void inv3(const double* a, double* o) {
    double M[9];
    M[0] = a[4] * a[8] - a[5] * a[7]; M[1] = a[2] * a[7] - a[1] * a[8]; M[2] = a[1] * a[5] - a[2] * a[4];
    M[3] = a[5] * a[6] - a[3] * a[8]; M[4] = a[0] * a[8] - a[2] * a[6]; M[5] = a[2] * a[3] - a[0] * a[5];
    M[6] = a[3] * a[7] - a[4] * a[6]; M[7] = a[1] * a[6] - a[0] * a[7]; M[8] = a[0] * a[4] - a[1] * a[3];
    const double idet = 1.0 / (a[0] * M[0] + a[1] * M[3] + a[2] * M[6]);
#pragma unroll
    for (int i = 0; i < 9; ++i) o[i] = M[i] * idet;
}

DEV void rot_to_quat(const double* R, double* q) {
    const double tr = R[0] + R[4] + R[8] + 1.0;
    double S;
    if (tr > 1e-7) {
        S = 0.5 / sqrt(tr);
        q[0] = 0.25 / S;
        q[1] = (R[7] - R[5]) * S; q[2] = (R[2] - R[6]) * S; q[3] = (R[3] - R[1]) * S;
    } else if (R[0] > R[4] && R[0] > R[8]) {
        S = sqrt(1.0 + R[0] - R[4] - R[8]) * 2.0;
        q[0] = (R[7] - R[5]) / S; q[1] = 0.25 * S; q[2] = (R[3] + R[1]) / S; q[3] = (R[2] + R[6]) / S;
    } else if (R[4] > R[8]) {
        S = sqrt(1.0 + R[4] - R[0] - R[8]) * 2.0;
        q[0] = (R[2] - R[6]) / S; q[1] = (R[3] + R[1]) / S; q[2] = 0.25 * S; q[3] = (R[7] + R[5]) / S;
    } else {
        S = sqrt(1.0 + R[8] - R[0] - R[4]) * 2.0;
        q[0] = (R[3] - R[1]) / S; q[1] = (R[2] + R[6]) / S; q[2] = (R[7] + R[5]) / S; q[3] = 0.25 * S;
    }
}

DEV void quat_to_rot(const double* q, double* R) {
    const double aa = q[0] * q[0], ab = q[0] * q[1], ac = q[0] * q[2], ad = q[0] * q[3];
    const double bb = q[1] * q[1], bc = q[1] * q[2], bd = q[1] * q[3];
    const double cc = q[2] * q[2], cd = q[2] * q[3], dd = q[3] * q[3];
    R[0] = aa + bb - cc - dd; R[1] = 2.0 * (bc - ad);   R[2] = 2.0 * (ac + bd);
    R[3] = 2.0 * (ad + bc);   R[4] = aa - bb + cc - dd; R[5] = 2.0 * (cd - ab);
    R[6] = 2.0 * (bd - ac);   R[7] = 2.0 * (ab + cd);   R[8] = aa - bb - cc + dd;
}

DEV bool finite_all(const double* v, int n) {
    bool ok = true;
    for (int i = 0; i < n; ++i) ok = ok && !(isnan(v[i]) || isinf(v[i]));
    return ok;
}

// Lambda-Twist P3P; returns the number of candidate (R,t)
__device__ int p3p_lambdatwist(const double* y1_, const double* y2_, const double* y3_, const double* x1, const double* x2,
                               const double* x3, double (*Rs)[9], double (*Ts)[3]) {
    double y1[3] = {y1_[0], y1_[1], y1_[2]}, y2[3] = {y2_[0], y2_[1], y2_[2]}, y3[3] = {y3_[0], y3_[1], y3_[2]};
    normalize3(y1); normalize3(y2); normalize3(y3);
    const double b12 = -2.0 * dot3(y1, y2), b13 = -2.0 * dot3(y1, y3), b23 = -2.0 * dot3(y2, y3);
    const double d12[3] = {x1[0] - x2[0], x1[1] - x2[1], x1[2] - x2[2]};
    const double d13[3] = {x1[0] - x3[0], x1[1] - x3[1], x1[2] - x3[2]};
    const double d23[3] = {x2[0] - x3[0], x2[1] - x3[1], x2[2] - x3[2]};
    double d12xd13[3];
    cross3(d12, d13, d12xd13);
    const double a12 = dot3(d12, d12), a13 = dot3(d13, d13), a23 = dot3(d23, d23);
    const double c31 = -0.5 * b13, c23 = -0.5 * b23, c12 = -0.5 * b12;
    const double blob = c12 * c23 * c31 - 1.0;
    const double s31_squared = 1.0 - c31 * c31, s23_squared = 1.0 - c23 * c23, s12_squared = 1.0 - c12 * c12;
    double p3 = a13 * (a23 * s31_squared - a13 * s23_squared);
    double p2 = 2.0 * blob * a23 * a13 + a13 * (2.0 * a12 + a13) * s23_squared + a23 * (a23 - a12) * s31_squared;
    double p1 = a23 * (a13 - a23) * s12_squared - a12 * a12 * s23_squared - 2.0 * a12 * (blob * a23 + a13 * s23_squared);
    double p0 = a12 * (a12 * s23_squared - a23 * s12_squared);
    p3 = 1.0 / p3;
    p2 *= p3; p1 *= p3; p0 *= p3;
    const double g = cubick(p2, p1, p0);
    const double A00 = a23 * (1.0 - g), A01 = (a23 * b12) * 0.5, A02 = (a23 * b13 * g) * (-0.5);
    const double A11 = a23 - a12 + a13 * g, A12 = b23 * (a13 * g - a12) * 0.5, A22 = g * (a13 - a23) - a12;
    const double A[9] = {A00, A01, A02, A01, A11, A12, A02, A12, A22};
    double V[9], L[3];
    eigwithknown0(A, V, L);
    const double qq = -L[1] / L[0];
    const double v = sqrt(qq > 0 ? qq : 0.0);
    int valid = 0;
    double Ls[4][3];
    for (int sgn = 0; sgn < 2; ++sgn) {
        const double s = sgn == 0 ? v : -v;
        const double w2 = 1.0 / (s * V[1] - V[0]);
        const double w0 = (V[3] - s * V[4]) * w2;
        const double w1 = (V[6] - s * V[7]) * w2;
        const double a = 1.0 / ((a13 - a12) * w1 * w1 - a12 * b13 * w1 - a12);
        const double b = (a13 * b12 * w1 - a12 * b13 * w0 - 2.0 * w0 * w1 * (a12 - a13)) * a;
        const double c = ((a13 - a12) * w0 * w0 + a13 * b12 * w0 + a13) * a;
        if (b * b - 4.0 * c >= 0) {
            double tau[2];
            root2real(b, c, tau[0], tau[1]);
            for (int k = 0; k < 2; ++k) {
                if (tau[k] > 0) {
                    const double t = tau[k];
                    const double d = a23 / (t * (b23 + t) + 1.0);
                    if (sgn == 1 && !(d > 0)) continue;
                    const double l2 = sqrt(d);
                    const double l3 = t * l2;
                    const double l1 = w0 * l2 + w1 * l3;
                    if (l1 >= 0 && valid < 4) { Ls[valid][0] = l1; Ls[valid][1] = l2; Ls[valid][2] = l3; ++valid; }
                }
            }
        }
    }
    for (int i = 0; i < valid; ++i) gauss_newton_refineL(Ls[i], a12, a13, a23, b12, b13, b23);
    const double X[9] = {d12[0], d13[0], d12xd13[0], d12[1], d13[1], d12xd13[1], d12[2], d13[2], d12xd13[2]};
    double Xi[9];
    inv3(X, Xi);
    for (int i = 0; i < valid; ++i) {
        double ry1[3], ry2[3], ry3[3], yd1[3], yd2[3], yx[3];
        for (int k = 0; k < 3; ++k) { ry1[k] = y1[k] * Ls[i][0]; ry2[k] = y2[k] * Ls[i][1]; ry3[k] = y3[k] * Ls[i][2]; }
        for (int k = 0; k < 3; ++k) { yd1[k] = ry1[k] - ry2[k]; yd2[k] = ry1[k] - ry3[k]; }
        cross3(yd1, yd2, yx);
        const double Y[9] = {yd1[0], yd2[0], yx[0], yd1[1], yd2[1], yx[1], yd1[2], yd2[2], yx[2]};
        for (int r = 0; r < 3; ++r)
            for (int cc = 0; cc < 3; ++cc) {
                double sacc = 0;
                for (int k = 0; k < 3; ++k) sacc += Y[r * 3 + k] * Xi[k * 3 + cc];
                Rs[i][r * 3 + cc] = sacc;
            }
        for (int r = 0; r < 3; ++r)
            Ts[i][r] = ry1[r] - (Rs[i][r * 3] * x1[0] + Rs[i][r * 3 + 1] * x1[1] + Rs[i][r * 3 + 2] * x1[2]);
    }
    return valid;
}

// p4p.cpp:11-60 -> pose (q,t); identity when no candidate survives
__device__ void p4p(const double* xs, const double* ys, const int* idx, double* q_out, double* t_out) {
    double Rs[4][9], Ts[4][3];
    double yh[3][3];
    for (int k = 0; k < 3; ++k) { yh[k][0] = ys[2 * idx[k]]; yh[k][1] = ys[2 * idx[k] + 1]; yh[k][2] = 1.0; }
    const int valid = p3p_lambdatwist(yh[0], yh[1], yh[2], xs + 3 * idx[0], xs + 3 * idx[1], xs + 3 * idx[2], Rs, Ts);
    const double* y = ys + 2 * idx[3];
    const double* x = xs + 3 * idx[3];
    q_out[0] = 1; q_out[1] = q_out[2] = q_out[3] = 0;
    t_out[0] = t_out[1] = t_out[2] = 0;
    double e0 = 1.7976931348623157e308;
    for (int v = 0; v < valid; ++v) {
        double q[4], R[9];
        rot_to_quat(Rs[v], q);
        const double ni = 1.0 / sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
        for (int k = 0; k < 4; ++k) q[k] *= ni;
        const double* t = Ts[v];
        if (!finite_all(q, 4) || !finite_all(t, 3)) continue;
        if (sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]) - 1.0 > 1e-5) continue;
        quat_to_rot(q, R);
        double xr[3];
        for (int r = 0; r < 3; ++r) xr[r] = R[3 * r] * x[0] + R[3 * r + 1] * x[1] + R[3 * r + 2] * x[2] + t[r];
        if (xr[2] < 0) continue;
        const double izr = 1.0 / xr[2];
        const double ex = xr[0] * izr - y[0], ey = xr[1] * izr - y[1];
        const double e = ex * ex + ey * ey;
        if (isnan(e)) continue;
        if (e < e0) {
            for (int k = 0; k < 4; ++k) q_out[k] = q[k];
            for (int k = 0; k < 3; ++k) t_out[k] = t[k];
            e0 = e;
        }
    }
}

DEV uint64_t mix64(uint64_t z) {
    z += 0x9E3779B97F4A7C15ULL;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}

// 4 distinct indices in [0,n), ascending (std::set order of get4RandomInRange0)
DEV void sample4(uint64_t seed, uint32_t iter, int n, int* idx) {
    int cnt = 0;
    const uint64_t key = mix64(seed ^ ((uint64_t)iter * 0xD1342543DE82EF95ULL));
    for (uint32_t d = 0; cnt < 4; ++d) {
        const uint64_t h = mix64(key + d);
        const int v = (int)(((h >> 32) * (uint64_t)n) >> 32);
        bool dup = false;
        for (int k = 0; k < cnt; ++k) dup = dup || (idx[k] == v);
        if (dup) continue;
        int pos = cnt++;
        while (pos > 0 && idx[pos - 1] > v) { idx[pos] = idx[pos - 1]; --pos; }
        idx[pos] = v;
    }
}

DEV unsigned count_inliers(const double* xs, const double* ys, int n, double thr2, const double* q, const double* t) {
    double R[9];
    quat_to_rot(q, R);
    unsigned inl = 0;
    for (int i = 0; i < n; ++i) {
        const double* X = xs + 3 * i;
        const double x = R[0] * X[0] + R[1] * X[1] + R[2] * X[2] + t[0];
        const double y = R[3] * X[0] + R[4] * X[1] + R[5] * X[2] + t[1];
        const double z = R[6] * X[0] + R[7] * X[1] + R[8] * X[2] + t[2];
        const double iz = 1.0 / z;
        if (iz < 0) continue;
        const double e1 = x * iz - ys[2 * i], e2 = y * iz - ys[2 * i + 1];
        const double err = e1 * e1 + e2 * e2;
        inl += (err < thr2) ? 1u : 0u;
    }
    return inl;
}

// Wave-wide sum, the same value in every lane: DPP register moves inside a row of 16 lanes (quad permutes, half-row and row mirrors), the four
// row sums by v_readlane, added in row order (as csrc/lm_device.h: wsum).  The __shfl_xor butterfly this replaces is two ds_bpermute_b32
// per step on a double, ~120 cycles of dependent latency each: 27 sums x 6 steps per refinement iteration were most of PNP::refine's time.
template <int CTRL>
DEV double pnp_dpp(double v) {
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
DEV double wave_sum_d(double v) {
    v += pnp_dpp<0xB1>(v);            // lane ^ 1
    v += pnp_dpp<0x4E>(v);            // lane ^ 2
    v += pnp_dpp<0x141>(v);           // row_half_mirror
    v += pnp_dpp<0x140>(v);           // row_mirror -> every lane of a row holds the row sum
    const int lo = __double2loint(v), hi = __double2hiint(v);
    const double r0 = __hiloint2double(__builtin_amdgcn_readlane(hi, 0), __builtin_amdgcn_readlane(lo, 0));
    const double r1 = __hiloint2double(__builtin_amdgcn_readlane(hi, 16), __builtin_amdgcn_readlane(lo, 16));
    const double r2 = __hiloint2double(__builtin_amdgcn_readlane(hi, 32), __builtin_amdgcn_readlane(lo, 32));
    const double r3 = __hiloint2double(__builtin_amdgcn_readlane(hi, 48), __builtin_amdgcn_readlane(lo, 48));
    return ((r0 + r1) + r2) + r3;
}
DEV double bcast_d(double v, int src) { return __shfl(v, src, 64); }

DEV void quat_mul(const double* a, const double* b, double* o) {
    o[0] = a[0] * b[0] - a[1] * b[1] - a[2] * b[2] - a[3] * b[3];
    o[1] = a[0] * b[1] + a[1] * b[0] + a[2] * b[3] - a[3] * b[2];
    o[2] = a[0] * b[2] - a[1] * b[3] + a[2] * b[0] + a[3] * b[1];
    o[3] = a[0] * b[3] + a[1] * b[2] - a[2] * b[1] + a[3] * b[0];
}
DEV void quat_plus(const double* q, const double* d, double* o) {
    const double n = sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
    if (n > 0.0) {
        const double s = sin(n) / n;
        const double dq[4] = {cos(n), s * d[0], s * d[1], s * d[2]};
        quat_mul(dq, q, o);
    } else { o[0] = q[0]; o[1] = q[1]; o[2] = q[2]; o[3] = q[3]; }
}

// 6x6 Cholesky solve, fully unrolled with compile-time indices (as a function with runtime loops the factor lived in scratch memory),
// one reciprocal per pivot
DEV bool chol6(const double* A, const double* b, double* x) {
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
                if (!(s > 0)) ok = false;
                L[i][i] = sqrt(ok ? s : 1.0);
                dinv[i] = 1.0 / L[i][i];
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

constexpr int PNP_MAX_PER_LANE = 16;   // n <= 1024 points per object

// cost over the selected points; lanes stride over points; `sel` bit j = point lane + 64*j selected
DEV double refine_cost(const double* xs, const double* ys, int n, unsigned sel, int lane, const double* q, const double* t) {
    double R[9], c = 0;
    quat_to_rot(q, R);
    for (int j = 0; j < PNP_MAX_PER_LANE; ++j) {
        const int i = lane + 64 * j;
        if (i < n && ((sel >> j) & 1u)) {
            const double* X = xs + 3 * i;
            const double x = R[0] * X[0] + R[1] * X[1] + R[2] * X[2] + t[0];
            const double y = R[3] * X[0] + R[4] * X[1] + R[5] * X[2] + t[1];
            const double z = R[6] * X[0] + R[7] * X[1] + R[8] * X[2] + t[2];
            const double iz = 1.0 / z;
            const double r0 = x * iz - ys[2 * i], r1 = y * iz - ys[2 * i + 1];
            c += r0 * r0 + r1 * r1;
        }
    }
    return 0.5 * wave_sum_d(c);
}

__device__ void refine_pass(const double* xs, const double* ys, int n, unsigned sel, int lane, double* q, double* t,
                            int max_iter, double tol) {
    double radius = 1e4, decrease = 2.0;
    double cost = refine_cost(xs, ys, n, sel, lane, q, t);
    for (int it = 0; it < max_iter; ++it) {
        double H[36], g[6], R[9];
        for (int a = 0; a < 36; ++a) H[a] = 0;
        for (int a = 0; a < 6; ++a) g[a] = 0;
        quat_to_rot(q, R);
        for (int j = 0; j < PNP_MAX_PER_LANE; ++j) {
            const int i = lane + 64 * j;
            if (i < n && ((sel >> j) & 1u)) {
                const double* X = xs + 3 * i;
                const double rx = R[0] * X[0] + R[1] * X[1] + R[2] * X[2];
                const double ry = R[3] * X[0] + R[4] * X[1] + R[5] * X[2];
                const double rz = R[6] * X[0] + R[7] * X[1] + R[8] * X[2];
                const double x = rx + t[0], y = ry + t[1], z = rz + t[2];
                const double iz = 1.0 / z;
                const double r[2] = {x * iz - ys[2 * i], y * iz - ys[2 * i + 1]};
                const double P[2][3] = {{iz, 0, -x * iz * iz}, {0, iz, -y * iz * iz}};
                const double D[3][6] = {{0, 2 * rz, -2 * ry, 1, 0, 0}, {-2 * rz, 0, 2 * rx, 0, 1, 0}, {2 * ry, -2 * rx, 0, 0, 0, 1}};
                double J[2][6];
                for (int a = 0; a < 2; ++a)
                    for (int c = 0; c < 6; ++c) J[a][c] = P[a][0] * D[0][c] + P[a][1] * D[1][c] + P[a][2] * D[2][c];
                for (int a = 0; a < 6; ++a) {
                    g[a] += J[0][a] * r[0] + J[1][a] * r[1];
                    for (int c = a; c < 6; ++c) H[a * 6 + c] += J[0][a] * J[0][c] + J[1][a] * J[1][c];
                }
            }
        }
        for (int a = 0; a < 6; ++a) {
            g[a] = wave_sum_d(g[a]);
            for (int c = a; c < 6; ++c) { H[a * 6 + c] = wave_sum_d(H[a * 6 + c]); H[c * 6 + a] = H[a * 6 + c]; }
        }
        double gmax = 0;
        for (int a = 0; a < 6; ++a) gmax = fmax(gmax, fabs(g[a]));
        if (gmax <= tol) return;
        double A[36], rhs[6], step[6];
        for (int a = 0; a < 36; ++a) A[a] = H[a];
        for (int a = 0; a < 6; ++a) {
            const double d = fmin(fmax(H[a * 6 + a], 1e-12), 1e64);
            A[a * 6 + a] += d / radius;
            rhs[a] = -g[a];
        }
        const bool ok = chol6(A, rhs, step);
        double rho = -1, new_cost = cost, qn[4], tn[3];
        if (ok) {
            double model = 0;
            for (int a = 0; a < 6; ++a) {
                double Hs = 0;
                for (int c = 0; c < 6; ++c) Hs += H[a * 6 + c] * step[c];
                model += -step[a] * (g[a] + 0.5 * Hs);
            }
            quat_plus(q, step, qn);
            for (int a = 0; a < 3; ++a) tn[a] = t[a] + step[3 + a];
            new_cost = refine_cost(xs, ys, n, sel, lane, qn, tn);
            rho = model > 0 ? (cost - new_cost) / model : -1;
        }
        if (ok && rho > 1e-3 && isfinite(new_cost)) {
            double snorm = 0, xnorm = 0;
            for (int a = 0; a < 6; ++a) snorm += step[a] * step[a];
            for (int a = 0; a < 4; ++a) xnorm += q[a] * q[a];
            for (int a = 0; a < 3; ++a) xnorm += t[a] * t[a];
            const double dc = cost - new_cost;
            for (int a = 0; a < 4; ++a) q[a] = qn[a];
            for (int a = 0; a < 3; ++a) t[a] = tn[a];
            const bool done = (fabs(dc) <= tol * cost) || (sqrt(snorm) <= 1e-8 * (sqrt(xnorm) + 1e-8));
            cost = new_cost;
            const double f = 1.0 - (2.0 * rho - 1.0) * (2.0 * rho - 1.0) * (2.0 * rho - 1.0);
            radius = fmin(radius / fmax(1.0 / 3.0, f), 1e16);
            decrease = 2.0;
            if (done) return;
        } else {
            radius /= decrease;
            decrease *= 2.0;
            if (radius < 1e-32) return;
        }
    }
}

// inlier (re)selection of PNP::refine; returns the selection mask for this lane's points, the
// selected count and (when prev is given) the number of flags that changed
DEV unsigned select_inliers(const double* xs, const double* ys, int n, double thr2, const double* q, const double* t, int lane,
                            const unsigned* prev, int* m_out, int* deltas_out) {
    double R[9];
    quat_to_rot(q, R);
    unsigned sel = 0;
    int m = 0, dl = 0;
    for (int j = 0; j < PNP_MAX_PER_LANE; ++j) {
        const int i = lane + 64 * j;
        if (i < n) {
            const double* X = xs + 3 * i;
            const double x = R[0] * X[0] + R[1] * X[1] + R[2] * X[2] + t[0];
            const double y = R[3] * X[0] + R[4] * X[1] + R[5] * X[2] + t[1];
            const double z = R[6] * X[0] + R[7] * X[1] + R[8] * X[2] + t[2];
            bool inl = !(z < 0);
            const double izr = 1.0 / z;
            const double ex = x * izr - ys[2 * i], ey = y * izr - ys[2 * i + 1];
            if (ex * ex + ey * ey > thr2) inl = false;
            if (prev && (inl != (((*prev) >> j) & 1u))) dl++;
            if (inl) { sel |= 1u << j; m++; }
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { m += __shfl_xor(m, o, 64); dl += __shfl_xor(dl, o, 64); }
    *m_out = m;
    if (deltas_out) *deltas_out = dl;
    return sel;
}

// ------------------------------------------------------------------------------------------------
// counts: nullptr = object o's points are [offsets[o], offsets[o + 1]); else [offsets[o], offsets[o] + counts[o]) -- problems compacted
// ON THE DEVICE into fixed-stride slots (csrc/frame_geom.hip), whose sizes the host never sees
// group_first (with counts): first problem of the group (frame) problem o belongs to; sampler keys then continue from group to group the
// way a caller advances its seed between per-frame launches (by the number of solvable problems of the frame)
//
// One WORKGROUP of PNP_WAVES waves per object.  The RANSAC loop is sequential by definition (PNP::compute accepts hypothesis i only if it
// beats everything before it, and every acceptance shortens the loop), but hypothesis i itself depends on i alone (counter-based
// sampler): the waves evaluate PNP_WAVES x 64 consecutive hypotheses at once, the inlier counts meet in LDS and every wave replays the
// sequential accept rule over them in index order -- hypotheses beyond the (shrinking) iteration count are discarded exactly as the
// sequential loop would never have drawn them.  1000 iterations (the cap: bad keypoints) are 4 rounds instead of 16, the usual 100-500
// are 1-2.  The refinement runs on wave 0.
// PNP_WAVES = 4 for launches of many objects (batched frames: more workgroups per CU), 16 for a frame or two (the one-frame call: 1000
// iterations -- the cap, what random-weight keypoints run to -- in ONE round instead of four: 152 -> ~70 us for 8 objects).  The result
// does not depend on it: the accept rule is replayed in index order whatever the round size.
// REPLAY (suo_pnp_replay: index-work parity with the reference's own sampler): hypothesis i of object o takes the 4-point sample draws[(o * n_draws + i) * 4 ..]
// -- a host-made table, e.g. the sequence std::default_random_engine + get4RandomInRange0 give (pnp_ransac.cpp:161-183) -- instead of sample4; win_out[o] = the
// hypothesis that became best_pose (-1: none).  Everything else is the same code.
template <int PNP_WAVES, bool REPLAY = false>
__global__ __launch_bounds__(64 * PNP_WAVES) void pnp_batch_kernel(const int* __restrict__ offsets, const int* __restrict__ counts,
                                                                   const int* __restrict__ group_first, const double* __restrict__ xs_all,
                                                                   const double* __restrict__ ys_all, double threshold, uint64_t seed,
                                                                   const int* __restrict__ iter_tab, const int* __restrict__ iter_tab_off,
                                                                   int do_refine, double* __restrict__ T_out, int* __restrict__ status,
                                                                   int* __restrict__ best_out, int* __restrict__ iters_out,
                                                                   const int* __restrict__ draws = nullptr, int n_draws = 0, int* __restrict__ win_out = nullptr,
                                                                   const uint64_t* __restrict__ seed_add = nullptr) {
    const int o = blockIdx.x, lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    __shared__ unsigned s_cnt[64 * PNP_WAVES];
    __shared__ double s_pose[2][8];
    const int p0 = offsets[o], n = counts ? counts[o] : offsets[o + 1] - p0;
    const double* xs = xs_all + 3 * (size_t)p0;
    const double* ys = ys_all + 2 * (size_t)p0;
    const int* tab = iter_tab + iter_tab_off[o];          // tab[b] = get_iterations(b / n), b = 0..n
    double bq[4] = {1, 0, 0, 0}, bt[3] = {0, 0, 0};
    unsigned best = 0;
    unsigned i_done = 0;
    int win_abs = -1;
    const bool solvable = n >= 4 && n <= 64 * PNP_MAX_PER_LANE;
    const double thr2 = threshold * threshold;
    if (solvable) {
        unsigned iters = (unsigned)tab[0];
        // sampler key: the object's rank among the launch's solvable problems (>= 4 points).  Host-compacted launches hold only
        // those (rank = o); device-compacted ones (counts) keep a slot per crop, so the rank is counted here -- the same object then
        // draws the same samples on either route (suo_slam_amd/object_slam.py: the host route skips objects with < 4 keypoints)
        int rank = o, before = 0;
        if (counts) {
            const int g0 = group_first ? group_first[o] : 0;
            int r = 0, b = 0;
            for (int i = lane; i < o; i += 64) {
                const int sv = counts[i] >= 4 ? 1 : 0;
                if (i < g0) b += sv; else r += sv;
            }
#pragma unroll
            for (int m = 32; m > 0; m >>= 1) { r += __shfl_xor(r, m, 64); b += __shfl_xor(b, m, 64); }
            rank = r; before = b;
        }
        // seed_add: the caller's running key lives on the device (suo_frame_geom_params.seed_dev): launch k + 1 is enqueued before launch k's counts are known
        const uint64_t oseed = seed + (seed_add ? *seed_add : 0ULL) + (uint64_t)before + (uint64_t)rank * 0x9E3779B97F4A7C15ULL;
        int par = 0;
        for (unsigned base = 0; base < iters; base += 64 * PNP_WAVES) {
            const unsigned i = base + wv * 64 + lane;
            double q[4], t[3];
            unsigned cnt = 0;
            if (base + wv * 64 < iters) {                  // (wave-uniform: a wave whose 64 hypotheses all lie beyond the loop skips them)
                int idx[4];
                if constexpr (REPLAY) {
                    const int* d = draws + ((size_t)o * n_draws + (i < (unsigned)n_draws ? i : 0u)) * 4;      // (beyond the loop bound: never accepted, any sample)
                    for (int k = 0; k < 4; ++k) idx[k] = d[k];
                } else
                sample4(oseed, i, n, idx);
                p4p(xs, ys, idx, q, t);
                cnt = count_inliers(xs, ys, n, thr2, q, t);
            }
            s_cnt[wv * 64 + lane] = cnt;
            __syncthreads();
            // Replay the sequential accept rule of PNP::compute over these hypotheses, every wave identically:
            //     for j: if (base + j >= iters) break;  if (cnt[j] > best) { best = cnt[j]; win = j; iters = tab[best]; }
            // as a scan instead of a loop of 64 PNP_WAVES dependent LDS reads.  With P(j) = max(best, cnt[0 .. j-1]) the bound in force when the
            // loop reaches j is tab[P(j)]; tab is non-increasing, so "processed(j) = base + j < tab[P(j)]" holds for a prefix j < J.  The new
            // best is P(J), the winner the FIRST j < J that reaches it (if it exceeds the old best), the loop counter at exit base + J.
            int win = -1;
            {
                constexpr int E = PNP_WAVES;                    // entries per lane: j = lane * E + e
                unsigned c[E], run = 0;
#pragma unroll
                for (int e = 0; e < E; ++e) { c[e] = s_cnt[lane * E + e]; run = c[e] > run ? c[e] : run; }
                unsigned incl = run;                            // inclusive scan of the lane maxima
#pragma unroll
                for (int d = 1; d < 64; d <<= 1) {
                    const unsigned o = (unsigned)__shfl_up((int)incl, d, 64);
                    if (lane >= d) incl = o > incl ? o : incl;
                }
                unsigned P = (unsigned)__shfl_up((int)incl, 1, 64);
                P = lane == 0 ? best : (P > best ? P : best);   // P(j) of the lane's first entry
                int first_stop = 64 * E;                        // first entry the loop does not reach
                unsigned seen = 0;                              // maximum over the lane's processed entries
                int first_of[E];
#pragma unroll
                for (int e = 0; e < E; ++e) {
                    const int j = lane * E + e;
                    const bool processed = base + (unsigned)j < (unsigned)tab[P];
                    if (!processed && first_stop == 64 * E) first_stop = j;
                    if (processed && first_stop == 64 * E) seen = c[e] > seen ? c[e] : seen;
                    first_of[e] = j;
                    P = c[e] > P ? c[e] : P;
                }
#pragma unroll
                for (int m = 32; m > 0; m >>= 1) {
                    const int o = __shfl_xor(first_stop, m, 64);
                    first_stop = o < first_stop ? o : first_stop;
                }
                const int J = first_stop;
                // entries of lanes wholly beyond J do not count
                unsigned nb = lane * E < J ? seen : 0u;
#pragma unroll
                for (int m = 32; m > 0; m >>= 1) {
                    const unsigned o = (unsigned)__shfl_xor((int)nb, m, 64);
                    nb = o > nb ? o : nb;
                }
                if (nb > best) {
                    int w0 = 64 * E;
#pragma unroll
                    for (int e = E - 1; e >= 0; --e)
                        if (first_of[e] < J && c[e] == nb) w0 = first_of[e];
#pragma unroll
                    for (int m = 32; m > 0; m >>= 1) {
                        const int o = __shfl_xor(w0, m, 64);
                        w0 = o < w0 ? o : w0;
                    }
                    win = w0;
                    win_abs = (int)base + w0;
                    best = nb;
                    iters = (unsigned)tab[best];
                }
                if (J > 0) i_done = base + (unsigned)J;          // total_iters of PNP::compute = loop counter at exit
            }
            if (win >= 0) {
                if (wv == (win >> 6) && lane == (win & 63)) {
                    for (int k = 0; k < 4; ++k) s_pose[par][k] = q[k];
                    for (int k = 0; k < 3; ++k) s_pose[par][4 + k] = t[k];
                }
            }
            __syncthreads();                               // (also: every wave has read s_cnt before the next round overwrites it)
            if (win >= 0) {
                for (int k = 0; k < 4; ++k) bq[k] = s_pose[par][k];
                for (int k = 0; k < 3; ++k) bt[k] = s_pose[par][4 + k];
            }
            par ^= 1;
        }
    }
    if (wv != 0) return;
    if (solvable && best > 3 && do_refine) {
        int m = 0, deltas = 0;
        unsigned sel = select_inliers(xs, ys, n, thr2, bq, bt, lane, nullptr, &m, nullptr);
        refine_pass(xs, ys, n, sel, lane, bq, bt, 5, 1e-6);
        const unsigned prev = sel;
        sel = select_inliers(xs, ys, n, thr2, bq, bt, lane, &prev, &m, &deltas);
        if (!((double)deltas < 0.05 * (double)m)) refine_pass(xs, ys, n, sel, lane, bq, bt, 3, 1e-8);
    }
    if (lane == 0) {
        double R[9];
        quat_to_rot(bq, R);
        double* T = T_out + 16 * (size_t)o;
        for (int r = 0; r < 3; ++r) { for (int c = 0; c < 3; ++c) T[4 * r + c] = R[3 * r + c]; T[4 * r + 3] = bt[r]; }
        T[12] = T[13] = T[14] = 0; T[15] = 1;
        // identity pose == total failure (pnp_ransac.cpp:231; caller test at object_slam.py:38)
        const bool ident = bq[0] == 1 && bq[1] == 0 && bq[2] == 0 && bq[3] == 0 && bt[0] == 0 && bt[1] == 0 && bt[2] == 0;
        status[o] = ident ? 1 : 0;
        if (best_out) best_out[o] = (int)best;
        if (iters_out) iters_out[o] = (int)i_done;
        if (REPLAY && win_out) win_out[o] = win_abs;
    }
}

int launch_pnp_batch_counts(int n_obj, const int* offsets, const int* counts, const int* group_first, const double* xs, const double* ys, double threshold, uint64_t seed,
                            const int* iter_tab, const int* iter_tab_off, int do_refine, double* T_out, int* status, int* best_out,
                            int* iters_out, hipStream_t s, const uint64_t* seed_add = nullptr) {
    if (n_obj <= 0) return SUO_OK;
    static const int wide_upto = (int)SUO_TUNE("SUO_PNP_WIDE_UPTO", 32);      // objects per launch that still take 16 waves each (0: never)
    if (n_obj <= wide_upto)
        hipLaunchKernelGGL(pnp_batch_kernel<16>, dim3(n_obj), dim3(1024), 0, s, offsets, counts, group_first, xs, ys, threshold, seed, iter_tab, iter_tab_off,
                           do_refine, T_out, status, best_out, iters_out, (const int*)nullptr, 0, (int*)nullptr, seed_add);
    else
        hipLaunchKernelGGL(pnp_batch_kernel<4>, dim3(n_obj), dim3(256), 0, s, offsets, counts, group_first, xs, ys, threshold, seed, iter_tab, iter_tab_off,
                           do_refine, T_out, status, best_out, iters_out, (const int*)nullptr, 0, (int*)nullptr, seed_add);
    SUO_HIP_CHECK(hipGetLastError());
    return SUO_OK;
}
// hypothesis samples from a table (draws [n_obj][n_draws][4], n_draws >= the iteration cap iter_tab[...][0]); win_out [n_obj]
int launch_pnp_replay(int n_obj, const int* offsets, const double* xs, const double* ys, double threshold, const int* iter_tab, const int* iter_tab_off,
                      int do_refine, const int* draws, int n_draws, double* T_out, int* status, int* best_out, int* iters_out, int* win_out, hipStream_t s) {
    if (n_obj <= 0) return SUO_OK;
    hipLaunchKernelGGL((pnp_batch_kernel<4, true>), dim3(n_obj), dim3(256), 0, s, offsets, (const int*)nullptr, (const int*)nullptr, xs, ys, threshold, (uint64_t)0, iter_tab,
                       iter_tab_off, do_refine, T_out, status, best_out, iters_out, draws, n_draws, win_out);
    SUO_HIP_CHECK(hipGetLastError());
    return SUO_OK;
}
int launch_pnp_batch(int n_obj, const int* offsets, const double* xs, const double* ys, double threshold, uint64_t seed,
                     const int* iter_tab, const int* iter_tab_off, int do_refine, double* T_out, int* status, int* best_out,
                     int* iters_out, hipStream_t s) {
    return launch_pnp_batch_counts(n_obj, offsets, nullptr, nullptr, xs, ys, threshold, seed, iter_tab, iter_tab_off, do_refine, T_out, status, best_out,
                                   iters_out, s, nullptr);
}

}  // namespace suo
