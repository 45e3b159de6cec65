/* ORACLE (test infrastructure only) -- CPU restatement, in plain C double precision, of the
 * reference's PnP path.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load the library built from this file; the product (suo_slam_amd) never does.
 *
 * Restates (paths relative to /root/reference/thirdparty/lambdatwist):
 *   p3p_lambdatwist<double,5>    lambdatwist/lambdatwist.p3p.h:33-339
 *   cubick / root2real           lambdatwist/solve_cubic.h:134-209, 14-34
 *   eigwithknown0                lambdatwist/solve_eig0.h:11-85
 *   gauss_newton_refineL<5>      lambdatwist/refine_lambda.h:21-102
 *   p4p                          p4p.cpp:11-60  (getRotationQuaternion utils/cvl/rotation_helpers.h:253-304,
 *                                               getRotationMatrix :213-241, Pose::isnormal utils/cvl/pose.h:381-386)
 *   evaluate_inlier_set          pnp_ransac.cpp:41-87
 *   PnpParams::get_iterations    parameters.h:76-102
 *   get4RandomInRange0           pnp_ransac.cpp:161-183   (4 distinct indices, ascending order)
 *   PNP::compute                 pnp_ransac.cpp:188-232
 *   PNP::refine                  pnp_ransac.cpp:240-326   (see below)
 *   py_pnp                       pnp_python_binding.cpp:32-54 (row-major 4x4 out, identity on failure)
 *
 * Pinning: P3P/P4P are checked against the reference's own p4p.cpp compiled from where it lies
 * (oracle/Makefile target `ref` -> oracle/_ref/libp4p_ref.so) and against the known-answer vector
 * of test_pnp.py:5-14 (tests/golden/pnp_test_vector.json).
 *
 * Two documented deviations, both "parity unpinned" at the reference:
 *  (1) Sampling.  The reference draws from a process-global libstdc++ minstd_rand0 whose state
 *      carries across calls (utils/random.h:66-116); that sequence is not portable and not a
 *      property of the algorithm.  Here hypothesis i draws from a counter-based generator
 *      keyed by (seed, i) -- the same one the HIP kernel uses -- so runs are reproducible and
 *      hypotheses are order-independent.  The sequential accept rule (strictly more inliers,
 *      adaptive iteration count) is restated exactly.
 *  (2) Refinement.  The reference calls system Ceres (version unpinned, absent from
 *      /root/reference and from this image: CMakeLists.txt:32).  Restated from Ceres' published
 *      Levenberg-Marquardt trust-region algorithm: residual x/z - y over (unit quaternion with
 *      the QuaternionParameterization 3-dof local update, translation), LM diagonal
 *      diag(J'J) clamped to [1e-6,1e32]^2 / radius, radius0 = 1e4, step accepted when
 *      rho > 1e-3, radius /= max(1/3, 1-(2rho-1)^3) on success, radius /= 2,4,8.. on failure,
 *      termination on function / gradient / parameter tolerance or the iteration cap
 *      (5 its tol 1e-6, then 3 its tol 1e-8 when >= 5% of the inlier set changed).
 */
#include <math.h>
#include <stdint.h>
#include <string.h>

#define ORC_API __attribute__((visibility("default")))

/* ---------------------------------------------------------------- small helpers */
static double dot3(const double* a, const double* b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
static void cross3(const double* a, const double* b, double* c) {
    c[0] = a[1] * b[2] - a[2] * b[1];
    c[1] = a[2] * b[0] - a[0] * b[2];
    c[2] = a[0] * b[1] - a[1] * b[0];
}
static void normalize3(double* a) {
    double si = 1.0 / sqrt(a[0] * a[0] + a[1] * a[1] + a[2] * a[2]);   /* Matrix::operator/= multiplies by 1/s (utils/cvl/matrix.h:401-409) */
    a[0] *= si; a[1] *= si; a[2] *= si;
}

/* x*x + b*x + c  (solve_cubic.h:14-34) */
static int root2real(double b, double c, double* r1, double* r2) {
    double v = b * b - 4.0 * c;
    if (v < 0) { *r1 = *r2 = 0.5 * b; return 0; }
    double y = sqrt(v);
    if (b < 0) { *r1 = 0.5 * (-b + y); *r2 = 0.5 * (-b - y); }
    else { *r1 = 2.0 * c / (-b + y); *r2 = 2.0 * c / (-b - y); }
    return 1;
}

/* one real root of r^3 + b r^2 + c r + d (solve_cubic.h:134-209) */
static double cubick(double b, double c, double d) {
    double r0;
    if (b * b >= 3.0 * c) {
        double v = sqrt(b * b - 3.0 * c);
        double t1 = (-b - v) / 3.0;
        double k = ((t1 + b) * t1 + c) * t1 + d;
        if (k > 0.0) {
            r0 = t1 - sqrt(-k / (3.0 * t1 + b));
        } else {
            double t2 = (-b + v) / 3.0;
            k = ((t2 + b) * t2 + c) * t2 + d;
            r0 = t2 + sqrt(-k / (3.0 * t2 + b));
        }
    } else {
        r0 = -b / 3.0;
        if (fabs((3.0 * r0 + 2.0 * b) * r0 + c) < 1e-4) r0 += 1;
    }
    for (unsigned cnt = 0; cnt < 50; ++cnt) {
        double fx = ((r0 + b) * r0 + c) * r0 + d;
        if (cnt < 7 || fabs(fx) > 1e-13) {
            double fpx = (3.0 * r0 + 2.0 * b) * r0 + c;
            r0 -= fx / fpx;
        } else break;
    }
    return r0;
}

/* eigen-decomposition of a symmetric 3x3 with one zero eigenvalue (solve_eig0.h:11-85);
 * x and E row-major, E columns = eigenvectors (v1,v2,v3). */
static void eigwithknown0(const double* x, double* E, double* L) {
    L[2] = 0;
    double v3[3] = {x[3] * x[7] - x[6] * x[4], x[6] * x[1] - x[7] * x[0], x[4] * x[0] - x[3] * x[1]};
    normalize3(v3);
    double x01_squared = x[1] * x[1];
    double b = -x[0] - x[4] - x[8];
    double c = -x01_squared - x[2] * x[2] - x[5] * x[5] + x[0] * (x[4] + x[8]) + x[4] * x[8];
    double e1, e2;
    root2real(b, c, &e1, &e2);
    if (fabs(e1) < fabs(e2)) { double t = e1; e1 = e2; e2 = t; }
    L[0] = e1; L[1] = e2;
    double mx0011 = -x[0] * x[4];
    double prec_0 = x[1] * x[5] - x[2] * x[4];
    double prec_1 = x[1] * x[2] - x[0] * x[5];
    double e = e1;
    double tmp = 1.0 / (e * (x[0] + x[4]) + mx0011 - e * e + x01_squared);
    double a1 = -(e * x[2] + prec_0) * tmp;
    double a2 = -(e * x[5] + prec_1) * tmp;
    double rnorm = 1.0 / sqrt(a1 * a1 + a2 * a2 + 1.0);
    a1 *= rnorm; a2 *= rnorm;
    double tmp2 = 1.0 / (e2 * (x[0] + x[4]) + mx0011 - e2 * e2 + x01_squared);
    double a21 = -(e2 * x[2] + prec_0) * tmp2;
    double a22 = -(e2 * x[5] + prec_1) * tmp2;
    double rnorm2 = 1.0 / sqrt(a21 * a21 + a22 * a22 + 1.0);
    a21 *= rnorm2; a22 *= rnorm2;
    E[0] = a1;    E[1] = a21;    E[2] = v3[0];
    E[3] = a2;    E[4] = a22;    E[5] = v3[1];
    E[6] = rnorm; E[7] = rnorm2; E[8] = v3[2];
}

/* refine_lambda.h:21-102 with iterations = 5 */
static void gauss_newton_refineL(double* L, double a12, double a13, double a23, double b12, double b13, double b23) {
    for (int i = 0; i < 5; ++i) {
        double l1 = L[0], l2 = L[1], l3 = L[2];
        double r1 = l1 * l1 + l2 * l2 + b12 * l1 * l2 - a12;
        double r2 = l1 * l1 + l3 * l3 + b13 * l1 * l3 - a13;
        double r3 = l2 * l2 + l3 * l3 + b23 * l2 * l3 - a23;
        if (fabs(r1) + fabs(r2) + fabs(r3) < 1e-10) break;
        double v0 = 2.0 * l1 + b12 * l2, v1 = 2.0 * l2 + b12 * l1;
        double v3 = 2.0 * l1 + b13 * l3, v5 = 2.0 * l3 + b13 * l1;
        double v7 = 2.0 * l2 + b23 * l3, v8 = 2.0 * l3 + b23 * l2;
        double det = 1.0 / (-v0 * v5 * v7 - v1 * v3 * v8);
        double J0 = -v5 * v7, J1 = -v1 * v8, J2 = v1 * v5;
        double J3 = -v3 * v8, J4 = v0 * v8, J5 = -v0 * v5;
        double J6 = v3 * v7, J7 = -v0 * v7, J8 = -v1 * v3;
        double n1 = l1 - det * (J0 * r1 + J1 * r2 + J2 * r3);
        double n2 = l2 - det * (J3 * r1 + J4 * r2 + J5 * r3);
        double n3 = l3 - det * (J6 * r1 + J7 * r2 + J8 * r3);
        double r11 = n1 * n1 + n2 * n2 + b12 * n1 * n2 - a12;
        double r12 = n1 * n1 + n3 * n3 + b13 * n1 * n3 - a13;
        double r13 = n2 * n2 + n3 * n3 + b23 * n2 * n3 - a23;
        if (fabs(r11) + fabs(r12) + fabs(r13) > fabs(r1) + fabs(r2) + fabs(r3)) break;
        L[0] = n1; L[1] = n2; L[2] = n3;
    }
}

static void inv3(const double* a, double* o) {   /* utils/cvl/matrix.h:633-652 */
    double M[9];
    M[0] = a[4] * a[8] - a[5] * a[7]; M[1] = a[2] * a[7] - a[1] * a[8]; M[2] = a[1] * a[5] - a[2] * a[4];
    M[3] = a[5] * a[6] - a[3] * a[8]; M[4] = a[0] * a[8] - a[2] * a[6]; M[5] = a[2] * a[3] - a[0] * a[5];
    M[6] = a[3] * a[7] - a[4] * a[6]; M[7] = a[1] * a[6] - a[0] * a[7]; M[8] = a[0] * a[4] - a[1] * a[3];
    double idet = 1.0 / (a[0] * M[0] + a[1] * M[3] + a[2] * M[6]);
    for (int i = 0; i < 9; ++i) o[i] = M[i] * idet;
}

/* lambdatwist.p3p.h:33-339.  y: un-normalised bearings (homogeneous image points); Rs row-major */
ORC_API int orc_p3p(const double* y1_, const double* y2_, const double* y3_, const double* x1, const double* x2,
                    const double* x3, double* Rs, double* Ts) {
    double y1[3] = {y1_[0], y1_[1], y1_[2]}, y2[3] = {y2_[0], y2_[1], y2_[2]}, y3[3] = {y3_[0], y3_[1], y3_[2]};
    normalize3(y1); normalize3(y2); normalize3(y3);
    double b12 = -2.0 * dot3(y1, y2), b13 = -2.0 * dot3(y1, y3), b23 = -2.0 * dot3(y2, y3);
    double d12[3] = {x1[0] - x2[0], x1[1] - x2[1], x1[2] - x2[2]};
    double d13[3] = {x1[0] - x3[0], x1[1] - x3[1], x1[2] - x3[2]};
    double d23[3] = {x2[0] - x3[0], x2[1] - x3[1], x2[2] - x3[2]};
    double d12xd13[3];
    cross3(d12, d13, d12xd13);
    double a12 = dot3(d12, d12), a13 = dot3(d13, d13), a23 = dot3(d23, d23);
    double c31 = -0.5 * b13, c23 = -0.5 * b23, c12 = -0.5 * b12;
    double blob = c12 * c23 * c31 - 1.0;
    double s31_squared = 1.0 - c31 * c31, s23_squared = 1.0 - c23 * c23, s12_squared = 1.0 - c12 * c12;
    double p3 = a13 * (a23 * s31_squared - a13 * s23_squared);
    double p2 = 2.0 * blob * a23 * a13 + a13 * (2.0 * a12 + a13) * s23_squared + a23 * (a23 - a12) * s31_squared;
    double p1 = a23 * (a13 - a23) * s12_squared - a12 * a12 * s23_squared - 2.0 * a12 * (blob * a23 + a13 * s23_squared);
    double p0 = a12 * (a12 * s23_squared - a23 * s12_squared);
    p3 = 1.0 / p3;
    p2 *= p3; p1 *= p3; p0 *= p3;
    double g = cubick(p2, p1, p0);
    double A00 = a23 * (1.0 - g), A01 = (a23 * b12) * 0.5, A02 = (a23 * b13 * g) * (-0.5);
    double A11 = a23 - a12 + a13 * g, A12 = b23 * (a13 * g - a12) * 0.5, A22 = g * (a13 - a23) - a12;
    double A[9] = {A00, A01, A02, A01, A11, A12, A02, A12, A22};
    double V[9], L[3];
    eigwithknown0(A, V, L);
    double q = -L[1] / L[0];
    double v = sqrt(q > 0 ? q : 0.0);
    int valid = 0;
    double Ls[4][3];
    for (int sgn = 0; sgn < 2; ++sgn) {
        double s = sgn == 0 ? v : -v;
        double w2 = 1.0 / (s * V[1] - V[0]);
        double w0 = (V[3] - s * V[4]) * w2;
        double w1 = (V[6] - s * V[7]) * w2;
        double a = 1.0 / ((a13 - a12) * w1 * w1 - a12 * b13 * w1 - a12);
        double b = (a13 * b12 * w1 - a12 * b13 * w0 - 2.0 * w0 * w1 * (a12 - a13)) * a;
        double c = ((a13 - a12) * w0 * w0 + a13 * b12 * w0 + a13) * a;
        if (b * b - 4.0 * c >= 0) {
            double tau[2];
            root2real(b, c, &tau[0], &tau[1]);
            for (int k = 0; k < 2; ++k) {
                if (tau[k] > 0) {
                    double t = tau[k];
                    double d = a23 / (t * (b23 + t) + 1.0);
                    if (sgn == 1 && !(d > 0)) continue;      /* the -v branch guards d>0 (:251,:265) */
                    double l2 = sqrt(d);
                    double l3 = t * l2;
                    double l1 = w0 * l2 + w1 * l3;
                    if (l1 >= 0) { Ls[valid][0] = l1; Ls[valid][1] = l2; Ls[valid][2] = l3; ++valid; }
                }
            }
        }
    }
    for (int i = 0; i < valid; ++i) gauss_newton_refineL(Ls[i], a12, a13, a23, b12, b13, b23);
    double X[9] = {d12[0], d13[0], d12xd13[0], d12[1], d13[1], d12xd13[1], d12[2], d13[2], d12xd13[2]};
    double Xi[9];
    inv3(X, Xi);
    for (int i = 0; i < valid; ++i) {
        double ry1[3], ry2[3], ry3[3], yd1[3], yd2[3], yx[3];
        for (int k = 0; k < 3; ++k) { ry1[k] = y1[k] * Ls[i][0]; ry2[k] = y2[k] * Ls[i][1]; ry3[k] = y3[k] * Ls[i][2]; }
        for (int k = 0; k < 3; ++k) { yd1[k] = ry1[k] - ry2[k]; yd2[k] = ry1[k] - ry3[k]; }
        cross3(yd1, yd2, yx);
        double Y[9] = {yd1[0], yd2[0], yx[0], yd1[1], yd2[1], yx[1], yd1[2], yd2[2], yx[2]};
        double* R = Rs + 9 * i;
        for (int r = 0; r < 3; ++r)
            for (int cc = 0; cc < 3; ++cc) {
                double sacc = 0;
                for (int k = 0; k < 3; ++k) sacc += Y[r * 3 + k] * Xi[k * 3 + cc];
                R[r * 3 + cc] = sacc;
            }
        for (int r = 0; r < 3; ++r)
            Ts[3 * i + r] = ry1[r] - (R[r * 3] * x1[0] + R[r * 3 + 1] * x1[1] + R[r * 3 + 2] * x1[2]);
    }
    return valid;
}

/* utils/cvl/rotation_helpers.h:253-304 */
static void rot_to_quat(const double* R, double* q) {
    double tr = R[0] + R[4] + R[8] + 1.0, S;
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

/* utils/cvl/rotation_helpers.h:213-241 (no normalisation of q) */
static void quat_to_rot(const double* q, double* R) {
    double aa = q[0] * q[0], ab = q[0] * q[1], ac = q[0] * q[2], ad = q[0] * q[3];
    double bb = q[1] * q[1], bc = q[1] * q[2], bd = q[1] * q[3];
    double cc = q[2] * q[2], cd = q[2] * q[3], dd = q[3] * q[3];
    R[0] = aa + bb - cc - dd; R[1] = 2.0 * (bc - ad);   R[2] = 2.0 * (ac + bd);
    R[3] = 2.0 * (ad + bc);   R[4] = aa - bb + cc - dd; R[5] = 2.0 * (cd - ab);
    R[6] = 2.0 * (bd - ac);   R[7] = 2.0 * (ab + cd);   R[8] = aa - bb - cc + dd;
}

static int finite_all(const double* v, int n) {
    for (int i = 0; i < n; ++i) if (isnan(v[i]) || isinf(v[i])) return 0;
    return 1;
}

/* p4p.cpp:11-60; pose out as (q[4], t[3]); identity when no candidate survives */
ORC_API void orc_p4p(const double* xs, const double* ys, const int* idx, double* q_out, double* t_out) {
    double Rs[36], Ts[12];
    double yh[3][3];
    for (int k = 0; k < 3; ++k) { yh[k][0] = ys[2 * idx[k]]; yh[k][1] = ys[2 * idx[k] + 1]; yh[k][2] = 1.0; }
    int valid = orc_p3p(yh[0], yh[1], yh[2], xs + 3 * idx[0], xs + 3 * idx[1], xs + 3 * idx[2], Rs, Ts);
    const double* y = ys + 2 * idx[3];
    const double* x = xs + 3 * idx[3];
    q_out[0] = 1; q_out[1] = q_out[2] = q_out[3] = 0;
    t_out[0] = t_out[1] = t_out[2] = 0;
    double e0 = 1.7976931348623157e308;
    for (int v = 0; v < valid; ++v) {
        double q[4], R[9];
        rot_to_quat(Rs + 9 * v, q);
        double ni = 1.0 / sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
        for (int k = 0; k < 4; ++k) q[k] *= ni;
        const double* t = Ts + 3 * v;
        if (!finite_all(q, 4) || !finite_all(t, 3)) continue;
        if (sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]) - 1.0 > 1e-5) continue;
        quat_to_rot(q, R);
        double xr[3];
        for (int r = 0; r < 3; ++r) xr[r] = R[3 * r] * x[0] + R[3 * r + 1] * x[1] + R[3 * r + 2] * x[2] + t[r];
        if (xr[2] < 0) continue;
        double izr = 1.0 / xr[2];                                   /* dehom() = operator/ = times reciprocal */
        double ex = xr[0] * izr - y[0], ey = xr[1] * izr - y[1];
        double e = ex * ex + ey * ey;
        if (isnan(e)) continue;
        if (e < e0) {
            for (int k = 0; k < 4; ++k) q_out[k] = q[k];
            for (int k = 0; k < 3; ++k) t_out[k] = t[k];
            e0 = e;
        }
    }
}

/* parameters.h:76-102 */
ORC_API int orc_get_iterations(double estimated_inliers) {
    const double p_meets = 0.9, min_probability = 0.99999;
    const unsigned max_iterations = 1000, min_iterations = 100;
    double p_inlier = fmin(0.9, estimated_inliers * p_meets);
    p_inlier = fmin(fmax(p_inlier, 1e-2), 1 - 1e-8);
    if (p_inlier < 0.01) return (int)max_iterations;
    double p_failure = fmin(fmax(1.0 - min_probability, 1e-8), 0.01);
    double p_good = pow(p_inlier, 4);
    double iterations = ceil(log(p_failure) / log(1.0 - p_good)) + 50;
    if (iterations < min_iterations) return (int)min_iterations;
    if (iterations > max_iterations) return (int)max_iterations;
    return (int)iterations;
}

/* pnp_ransac.cpp:41-87; M = [R|t] from the pose; returns the exact count when best_inliers == 0 */
static unsigned evaluate_inlier_set(const double* xs, const double* ys, int n, double threshold, const double* q,
                                    const double* t, unsigned best_inliers) {
    double R[9];
    quat_to_rot(q, R);
    unsigned inliers = 0;
    double thr2 = threshold * threshold;
    for (int i = 0; i < n; ++i) {
        const double* X = xs + 3 * i;
        double x = R[0] * X[0] + R[1] * X[1] + R[2] * X[2] + t[0];
        double y = R[3] * X[0] + R[4] * X[1] + R[5] * X[2] + t[1];
        double z = R[6] * X[0] + R[7] * X[1] + R[8] * X[2] + t[2];
        double iz = 1.0 / z;
        if (iz < 0) continue;
        double e1 = x * iz - ys[2 * i], e2 = y * iz - ys[2 * i + 1];
        double err = e1 * e1 + e2 * e2;
        inliers += (err < thr2) ? 1 : 0;
        if ((unsigned)(n - i) + inliers < best_inliers) break;
    }
    return inliers;
}

/* counter-based sampler shared (by specification, not by code) with the HIP kernel */
static uint64_t mix64(uint64_t z) {
    z += 0x9E3779B97F4A7C15ULL;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}
ORC_API void orc_sample4(uint64_t seed, uint32_t iter, int n, int* idx) {
    int cnt = 0;
    uint64_t key = mix64(seed ^ ((uint64_t)iter * 0xD1342543DE82EF95ULL));
    for (uint32_t d = 0; cnt < 4; ++d) {
        uint64_t h = mix64(key + d);
        int v = (int)(((h >> 32) * (uint64_t)n) >> 32);
        int dup = 0;
        for (int k = 0; k < cnt; ++k) dup |= (idx[k] == v);
        if (dup) continue;
        int pos = cnt++;                       /* insert sorted ascending (std::set order) */
        while (pos > 0 && idx[pos - 1] > v) { idx[pos] = idx[pos - 1]; --pos; }
        idx[pos] = v;
    }
}

/* ---------------------------------------------------------------- refine (Ceres restatement) */
/* ---- the REFERENCE's own sampler (index-work parity) --------------------------------------------------------------
 * PNP::compute draws through get4RandomInRange0 (pnp_ransac.cpp:161-183): a std::set<uint> filled with mlib::randui<int>(0, max - 1)
 * (utils/random.h:112-116) until it holds 4 values, read out in ascending order.  randui = std::uniform_int_distribution<int>(lo, hi) over the
 * process-global std::default_random_engine seeded once with RANDOM_SEED_VALUE = 0 (random.h:40-42,65-86).  Both are libstdc++'s:
 *   default_random_engine = minstd_rand0: x <- 16807 x mod (2^31 - 1); seed 0 -> state 1; min 1, max 2^31 - 2
 *   uniform_int_distribution (range of the engine > range asked): scaling = urngrange / (hi - lo + 1), past = (hi - lo + 1) * scaling,
 *   draw x - 1 until it is < past, return lo + (x - 1) / scaling                         (bits/uniform_int_dist.h, the "downscaling" branch)
 * Pinned bit for bit against the reference's header compiled here (oracle/ref_random_shim.cpp -> tests/test_ref_sampler.py, tests/golden/sampler_golden.npz). */
ORC_API void orc_ref_rng_seed(uint32_t* state, uint32_t seed) {
    uint32_t x = (uint32_t)(seed % 2147483647u);
    *state = x == 0 ? 1u : x;
}
static uint32_t ref_rng_next(uint32_t* state) {
    *state = (uint32_t)(((uint64_t)*state * 16807u) % 2147483647u);
    return *state;
}
ORC_API int orc_ref_randui(uint32_t* state, int lo, int hi) {
    const uint64_t urngrange = 2147483646u - 1u, urange = (uint64_t)((unsigned)hi - (unsigned)lo);
    const uint64_t uerange = urange + 1, scaling = urngrange / uerange, past = uerange * scaling;
    uint64_t ret;
    do ret = (uint64_t)ref_rng_next(state) - 1u; while (ret >= past);
    return (int)(ret / scaling) + lo;
}
/* one get4RandomInRange0(max): 4 distinct values in [0, max), ascending; returns the number of randui calls it consumed */
ORC_API int orc_ref_get4(uint32_t* state, int max, int* idx) {
    int cnt = 0, calls = 0;
    while (cnt < 4) {
        const int v = orc_ref_randui(state, 0, max - 1);
        ++calls;
        int dup = 0;
        for (int k = 0; k < cnt; ++k) dup |= (idx[k] == v);
        if (dup) continue;
        int pos = cnt++;
        while (pos > 0 && idx[pos - 1] > v) { idx[pos] = idx[pos - 1]; --pos; }
        idx[pos] = v;
    }
    return calls;
}

static void quat_mul(const double* a, const double* b, double* o) {
    o[0] = a[0] * b[0] - a[1] * b[1] - a[2] * b[2] - a[3] * b[3];
    o[1] = a[0] * b[1] + a[1] * b[0] + a[2] * b[3] - a[3] * b[2];
    o[2] = a[0] * b[2] - a[1] * b[3] + a[2] * b[0] + a[3] * b[1];
    o[3] = a[0] * b[3] + a[1] * b[2] - a[2] * b[1] + a[3] * b[0];
}
/* ceres::QuaternionParameterization::Plus: x_plus = [cos|d|, sin|d|/|d| d] * x */
static void quat_plus(const double* q, const double* d, double* o) {
    double n = sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
    if (n > 0.0) {
        double s = sin(n) / n;
        double dq[4] = {cos(n), s * d[0], s * d[1], s * d[2]};
        quat_mul(dq, q, o);
    } else { o[0] = q[0]; o[1] = q[1]; o[2] = q[2]; o[3] = q[3]; }
}

static double refine_cost(const double* xs, const double* ys, const int* sel, int m, const double* q, const double* t) {
    double R[9], c = 0;
    quat_to_rot(q, R);
    for (int k = 0; k < m; ++k) {
        const double* X = xs + 3 * sel[k];
        double x = R[0] * X[0] + R[1] * X[1] + R[2] * X[2] + t[0];
        double y = R[3] * X[0] + R[4] * X[1] + R[5] * X[2] + t[1];
        double z = R[6] * X[0] + R[7] * X[1] + R[8] * X[2] + t[2];
        double iz = 1.0 / z;
        double r0 = x * iz - ys[2 * sel[k]], r1 = y * iz - ys[2 * sel[k] + 1];
        c += r0 * r0 + r1 * r1;
    }
    return 0.5 * c;
}

/* solve the symmetric positive definite 6x6 system A x = b by Cholesky; returns 0 on failure */
static int chol6(const double* A, const double* b, double* x) {
    double Lm[36];
    memset(Lm, 0, sizeof(Lm));
    for (int i = 0; i < 6; ++i)
        for (int j = 0; j <= i; ++j) {
            double s = A[i * 6 + j];
            for (int k = 0; k < j; ++k) s -= Lm[i * 6 + k] * Lm[j * 6 + k];
            if (i == j) { if (!(s > 0)) return 0; Lm[i * 6 + i] = sqrt(s); }
            else Lm[i * 6 + j] = s / Lm[j * 6 + j];
        }
    double y[6];
    for (int i = 0; i < 6; ++i) { double s = b[i]; for (int k = 0; k < i; ++k) s -= Lm[i * 6 + k] * y[k]; y[i] = s / Lm[i * 6 + i]; }
    for (int i = 5; i >= 0; --i) { double s = y[i]; for (int k = i + 1; k < 6; ++k) s -= Lm[k * 6 + i] * x[k]; x[i] = s / Lm[i * 6 + i]; }
    return 1;
}

static void refine_pass(const double* xs, const double* ys, const int* sel, int m, double* q, double* t, int max_iter, double tol) {
    double radius = 1e4, decrease = 2.0;
    double cost = refine_cost(xs, ys, sel, m, q, t);
    for (int it = 0; it < max_iter; ++it) {
        double H[36], g[6], R[9];
        memset(H, 0, sizeof(H));
        memset(g, 0, sizeof(g));
        quat_to_rot(q, R);
        for (int k = 0; k < m; ++k) {
            const double* X = xs + 3 * sel[k];
            double rx = R[0] * X[0] + R[1] * X[1] + R[2] * X[2];
            double ry = R[3] * X[0] + R[4] * X[1] + R[5] * X[2];
            double rz = R[6] * X[0] + R[7] * X[1] + R[8] * X[2];
            double x = rx + t[0], y = ry + t[1], z = rz + t[2];
            double iz = 1.0 / z;
            double r[2] = {x * iz - ys[2 * sel[k]], y * iz - ys[2 * sel[k] + 1]};
            /* d(proj)/dX */
            double P[2][3] = {{iz, 0, -x * iz * iz}, {0, iz, -y * iz * iz}};
            /* dX/d(delta) = -2 [R x]_x  (unit q, left perturbation of angle 2|delta|);  dX/dt = I */
            double D[3][6] = {{0, 2 * rz, -2 * ry, 1, 0, 0}, {-2 * rz, 0, 2 * rx, 0, 1, 0}, {2 * ry, -2 * rx, 0, 0, 0, 1}};
            double J[2][6];
            for (int a = 0; a < 2; ++a)
                for (int c = 0; c < 6; ++c) J[a][c] = P[a][0] * D[0][c] + P[a][1] * D[1][c] + P[a][2] * D[2][c];
            for (int a = 0; a < 6; ++a) {
                g[a] += J[0][a] * r[0] + J[1][a] * r[1];
                for (int c = 0; c < 6; ++c) H[a * 6 + c] += J[0][a] * J[0][c] + J[1][a] * J[1][c];
            }
        }
        double gmax = 0;
        for (int a = 0; a < 6; ++a) gmax = fmax(gmax, fabs(g[a]));
        if (gmax <= tol) return;                                   /* gradient_tolerance */
        double A[36], rhs[6], step[6];
        memcpy(A, H, sizeof(A));
        for (int a = 0; a < 6; ++a) {
            double d = fmin(fmax(H[a * 6 + a], 1e-12), 1e64);          /* (min,max)_lm_diagonal^2 */
            A[a * 6 + a] += d / radius;
            rhs[a] = -g[a];
        }
        int ok = chol6(A, rhs, step);
        double rho = -1, new_cost = cost, qn[4], tn[3];
        if (ok) {
            double Hs[6], model = 0;
            for (int a = 0; a < 6; ++a) { Hs[a] = 0; for (int c = 0; c < 6; ++c) Hs[a] += H[a * 6 + c] * step[c]; }
            for (int a = 0; a < 6; ++a) model += -step[a] * (g[a] + 0.5 * Hs[a]);   /* model cost decrease */
            quat_plus(q, step, qn);
            for (int a = 0; a < 3; ++a) tn[a] = t[a] + step[3 + a];
            new_cost = refine_cost(xs, ys, sel, m, qn, tn);
            rho = model > 0 ? (cost - new_cost) / model : -1;
        }
        if (ok && rho > 1e-3 && isfinite(new_cost)) {
            double snorm = 0, xnorm = 0;
            for (int a = 0; a < 6; ++a) snorm += step[a] * step[a];
            for (int a = 0; a < 4; ++a) xnorm += q[a] * q[a];
            for (int a = 0; a < 3; ++a) xnorm += t[a] * t[a];
            double dc = cost - new_cost;
            memcpy(q, qn, sizeof(qn));
            memcpy(t, tn, sizeof(tn));
            int done = (fabs(dc) <= tol * cost) || (sqrt(snorm) <= 1e-8 * (sqrt(xnorm) + 1e-8));
            cost = new_cost;
            double f = 1.0 - (2.0 * rho - 1.0) * (2.0 * rho - 1.0) * (2.0 * rho - 1.0);
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

static int select_inliers(const double* xs, const double* ys, int n, double thr2, const double* q, const double* t,
                          int* sel, int* flags, int* deltas) {
    double R[9];
    quat_to_rot(q, R);
    int m = 0, dl = 0;
    for (int i = 0; i < n; ++i) {
        const double* X = xs + 3 * i;
        double x = R[0] * X[0] + R[1] * X[1] + R[2] * X[2] + t[0];
        double y = R[3] * X[0] + R[4] * X[1] + R[5] * X[2] + t[1];
        double z = R[6] * X[0] + R[7] * X[1] + R[8] * X[2] + t[2];
        int inl = 1;
        if (z < 0) inl = 0;
        double izr = 1.0 / z;
        double ex = x * izr - ys[2 * i], ey = y * izr - ys[2 * i + 1];
        if (ex * ex + ey * ey > thr2) inl = 0;
        if (deltas && (inl ^ (flags[i] == 1))) dl++;
        if (!deltas) flags[i] = inl;
        if (inl) sel[m++] = i;
    }
    if (deltas) *deltas = dl;
    return m;
}

/* PNP::compute + refine + py_pnp output convention.  T_out row-major 4x4; returns number of RANSAC
 * iterations executed; *best_inliers_out the consensus size.  n >= 4 (caller guarantees, object_slam.py:31). */
/* draws != NULL: hypothesis i takes the sample draws[4 i .. 4 i + 3] (a table made by orc_ref_get4 -- the reference's sequence -- or anything else) instead of
 * the counter-based sampler; n_draws must cover the loop (the iteration cap, orc_get_iterations(0)).  *winner_out = index of the hypothesis that became best_pose. */
ORC_API int orc_pnp_ransac_draws(const double* xs, const double* ys, int n, double threshold, uint64_t seed, const int* draws, int n_draws, int do_refine,
                                 double* T_out, int* best_inliers_out, int* winner_out);
ORC_API int orc_pnp_ransac(const double* xs, const double* ys, int n, double threshold, uint64_t seed, int do_refine,
                           double* T_out, int* best_inliers_out) {
    return orc_pnp_ransac_draws(xs, ys, n, threshold, seed, 0, 0, do_refine, T_out, best_inliers_out, 0);
}
ORC_API int orc_pnp_ransac_draws(const double* xs, const double* ys, int n, double threshold, uint64_t seed, const int* draws, int n_draws, int do_refine,
                                 double* T_out, int* best_inliers_out, int* winner_out) {
    double bq[4] = {1, 0, 0, 0}, bt[3] = {0, 0, 0};
    unsigned best = 0;
    unsigned iters = (unsigned)orc_get_iterations(0.0);
    unsigned i;
    int winner = -1;
    for (i = 0; i < iters; ++i) {
        int idx[4];
        double q[4], t[3];
        if (draws) {
            if ((int)i >= n_draws) break;
            memcpy(idx, draws + 4 * i, sizeof(idx));
        } else
        orc_sample4(seed, i, n, idx);
        orc_p4p(xs, ys, idx, q, t);
        unsigned inl = evaluate_inlier_set(xs, ys, n, threshold, q, t, best);
        if (inl > best) {
            winner = (int)i;
            best = inl;
            memcpy(bq, q, sizeof(bq));
            memcpy(bt, t, sizeof(bt));
            iters = (unsigned)orc_get_iterations(best / (double)n);
        }
    }
    if (best > 3 && do_refine) {
        int sel[4096], flags[4096], deltas = 0;
        if (n <= 4096) {
            double thr2 = threshold * threshold;
            int m = select_inliers(xs, ys, n, thr2, bq, bt, sel, flags, 0);
            refine_pass(xs, ys, sel, m, bq, bt, 5, 1e-6);
            m = select_inliers(xs, ys, n, thr2, bq, bt, sel, flags, &deltas);
            if (!(deltas < 0.05 * m)) refine_pass(xs, ys, sel, m, bq, bt, 3, 1e-8);
        }
    }
    double R[9];
    quat_to_rot(bq, R);
    for (int r = 0; r < 3; ++r) { for (int c = 0; c < 3; ++c) T_out[4 * r + c] = R[3 * r + c]; T_out[4 * r + 3] = bt[r]; }
    T_out[12] = T_out[13] = T_out[14] = 0; T_out[15] = 1;
    if (best_inliers_out) *best_inliers_out = (int)best;
    if (winner_out) *winner_out = winner;
    return (int)i;
}
