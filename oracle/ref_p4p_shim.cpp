// Thin extern "C" shim over the REFERENCE's own P3P/P4P (compiled from where it lies under
// /root/reference/thirdparty/lambdatwist by `make -C oracle ref`; output oracle/_ref/libp4p_ref.so).
// Test infrastructure only: used to validate oracle/pnp_oracle.c and to generate golden vectors.
#include <vector>
#include <p4p.h>
#include <lambdatwist/lambdatwist.p3p.h>

extern "C" {

// xs [n,3], ys [n,2] normalised; idx[4]; out: row-major 4x4
void ref_p4p(const double* xs, const double* ys, int n, const int* idx, double* T16) {
    std::vector<cvl::Vector3D> X(n);
    std::vector<cvl::Vector2D> Y(n);
    for (int i = 0; i < n; ++i) {
        X[i] = cvl::Vector3D(xs[3 * i], xs[3 * i + 1], xs[3 * i + 2]);
        Y[i] = cvl::Vector2D(ys[2 * i], ys[2 * i + 1]);
    }
    cvl::Vector4<uint> I(idx[0], idx[1], idx[2], idx[3]);
    cvl::PoseD P = cvl::p4p(X, Y, I);
    cvl::Matrix4x4D M = P.get4x4();
    for (int r = 0; r < 4; ++r)
        for (int c = 0; c < 4; ++c) T16[4 * r + c] = M(r, c);
}

// raw Lambda-Twist: bearings y (homogeneous, un-normalised), points x; Rs [4][9] row-major, Ts [4][3]
int ref_p3p(const double* y1, const double* y2, const double* y3, const double* x1, const double* x2, const double* x3,
            double* Rs, double* Ts) {
    cvl::Vector<cvl::Matrix<double, 3, 3>, 4> R;
    cvl::Vector<cvl::Vector3<double>, 4> T;
    int valid = cvl::p3p_lambdatwist<double, 5>(cvl::Vector3D(y1[0], y1[1], y1[2]), cvl::Vector3D(y2[0], y2[1], y2[2]),
                                                cvl::Vector3D(y3[0], y3[1], y3[2]), cvl::Vector3D(x1[0], x1[1], x1[2]),
                                                cvl::Vector3D(x2[0], x2[1], x2[2]), cvl::Vector3D(x3[0], x3[1], x3[2]), R, T);
    for (int v = 0; v < valid; ++v) {
        for (int r = 0; r < 3; ++r)
            for (int c = 0; c < 3; ++c) Rs[9 * v + 3 * r + c] = R[v](r, c);
        for (int r = 0; r < 3; ++r) Ts[3 * v + r] = T[v][r];
    }
    return valid;
}
}
