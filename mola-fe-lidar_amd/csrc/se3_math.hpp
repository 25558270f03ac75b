// se3_math.hpp -- fp64 host-side geometry for the ICP core: pose conversions,
// SE(3) log / stall test, Horn's closed-form solve, 6x6 covariance.
//
// These are the O(1)-per-iteration pieces of the hot path (SURVEY.md §8 rows
// a9, a10, a12).  The reference reaches them through mp2p_icp::Solver_Horn and
// mp2p_icp::ICP::align (call site src/LidarOdometry.cpp:869-871); MRPT's
// TPose3D convention comes from src/LidarOdometry.cpp:272-275.
#pragma once
#include <cmath>
#include <cstring>

namespace mola_icp_amd {

struct Mat4 {
    double m[16];
    static Mat4 identity()
    {
        Mat4 r{};
        r.m[0] = r.m[5] = r.m[10] = r.m[15] = 1.0;
        return r;
    }
    double& operator()(int r, int c) { return m[4 * r + c]; }
    double operator()(int r, int c) const { return m[4 * r + c]; }
};

inline Mat4 mul(const Mat4& A, const Mat4& B)
{
    Mat4 C{};
    for (int i = 0; i < 4; ++i)
        for (int k = 0; k < 4; ++k) {
            const double a = A(i, k);
            for (int j = 0; j < 4; ++j) C(i, j) += a * B(k, j);
        }
    return C;
}

inline Mat4 inverse_rigid(const Mat4& T)
{
    Mat4 I = Mat4::identity();
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) I(i, j) = T(j, i);
    for (int i = 0; i < 3; ++i) I(i, 3) = -(I(i, 0) * T(0, 3) + I(i, 1) * T(1, 3) + I(i, 2) * T(2, 3));
    return I;
}

// MRPT TPose3D(x,y,z,yaw,pitch,roll): R = Rz(yaw) Ry(pitch) Rx(roll)
inline Mat4 pose_from_xyzypr(const double p[6])
{
    const double cy = std::cos(p[3]), sy = std::sin(p[3]);
    const double cp = std::cos(p[4]), sp = std::sin(p[4]);
    const double cr = std::cos(p[5]), sr = std::sin(p[5]);
    Mat4 T = Mat4::identity();
    T(0, 0) = cy * cp; T(0, 1) = cy * sp * sr - sy * cr; T(0, 2) = cy * sp * cr + sy * sr;
    T(1, 0) = sy * cp; T(1, 1) = sy * sp * sr + cy * cr; T(1, 2) = sy * sp * cr - cy * sr;
    T(2, 0) = -sp;     T(2, 1) = cp * sr;                T(2, 2) = cp * cr;
    T(0, 3) = p[0]; T(1, 3) = p[1]; T(2, 3) = p[2];
    return T;
}

inline void pose_to_xyzypr(const Mat4& T, double p[6])
{
    p[0] = T(0, 3); p[1] = T(1, 3); p[2] = T(2, 3);
    const double sp = -T(2, 0);
    if (std::fabs(sp) < 1.0 - 1e-12) {
        p[4] = std::asin(sp);
        p[3] = std::atan2(T(1, 0), T(0, 0));
        p[5] = std::atan2(T(2, 1), T(2, 2));
    } else {  // gimbal lock: put everything in yaw
        p[4] = sp > 0 ? M_PI / 2 : -M_PI / 2;
        p[3] = std::atan2(-T(0, 1), T(1, 1));
        p[5] = 0;
    }
}

// log: out = (v, w) with v = V(w)^-1 t, MRPT Lie::SE<3>::log ordering.
inline void se3_log(const Mat4& T, double out[6])
{
    // rotation part via the quaternion (stable for all angles)
    const double tr = T(0, 0) + T(1, 1) + T(2, 2);
    double qw, qx, qy, qz;
    if (tr > 0) {
        const double s = std::sqrt(tr + 1.0) * 2;
        qw = 0.25 * s; qx = (T(2, 1) - T(1, 2)) / s; qy = (T(0, 2) - T(2, 0)) / s; qz = (T(1, 0) - T(0, 1)) / s;
    } else if (T(0, 0) > T(1, 1) && T(0, 0) > T(2, 2)) {
        const double s = std::sqrt(1.0 + T(0, 0) - T(1, 1) - T(2, 2)) * 2;
        qw = (T(2, 1) - T(1, 2)) / s; qx = 0.25 * s; qy = (T(0, 1) + T(1, 0)) / s; qz = (T(0, 2) + T(2, 0)) / s;
    } else if (T(1, 1) > T(2, 2)) {
        const double s = std::sqrt(1.0 + T(1, 1) - T(0, 0) - T(2, 2)) * 2;
        qw = (T(0, 2) - T(2, 0)) / s; qx = (T(0, 1) + T(1, 0)) / s; qy = 0.25 * s; qz = (T(1, 2) + T(2, 1)) / s;
    } else {
        const double s = std::sqrt(1.0 + T(2, 2) - T(0, 0) - T(1, 1)) * 2;
        qw = (T(1, 0) - T(0, 1)) / s; qx = (T(0, 2) + T(2, 0)) / s; qy = (T(1, 2) + T(2, 1)) / s; qz = 0.25 * s;
    }
    if (qw < 0) { qw = -qw; qx = -qx; qy = -qy; qz = -qz; }
    const double vn = std::sqrt(qx * qx + qy * qy + qz * qz);
    double w[3];
    if (vn < 1e-12) {
        w[0] = 2 * qx; w[1] = 2 * qy; w[2] = 2 * qz;
    } else {
        const double ang = 2.0 * std::atan2(vn, qw);
        w[0] = ang * qx / vn; w[1] = ang * qy / vn; w[2] = ang * qz / vn;
    }
    const double th2 = w[0] * w[0] + w[1] * w[1] + w[2] * w[2];
    double c;  // V^-1 = I - 1/2 W + c W^2
    if (th2 < 1e-10) c = 1.0 / 12.0 + th2 / 720.0;
    else {
        const double th = std::sqrt(th2);
        c = (1.0 - 0.5 * th * std::sin(th) / (1.0 - std::cos(th))) / th2;
    }
    const double t[3] = {T(0, 3), T(1, 3), T(2, 3)};
    const double a[3] = {w[1] * t[2] - w[2] * t[1], w[2] * t[0] - w[0] * t[2], w[0] * t[1] - w[1] * t[0]};
    const double b[3] = {w[1] * a[2] - w[2] * a[1], w[2] * a[0] - w[0] * a[2], w[0] * a[1] - w[1] * a[0]};
    for (int i = 0; i < 3; ++i) {
        out[i] = t[i] - 0.5 * a[i] + c * b[i];
        out[3 + i] = w[i];
    }
}

// |v|, |w| of log(Tprev^-1 T): the stall-test quantities (icp-settings-regular.yaml:12-13)
inline void stall_deltas(const Mat4& T, const Mat4& Tprev, double& d_xyz, double& d_rot)
{
    double lg[6];
    se3_log(mul(inverse_rigid(Tprev), T), lg);
    d_xyz = std::sqrt(lg[0] * lg[0] + lg[1] * lg[1] + lg[2] * lg[2]);
    d_rot = std::sqrt(lg[3] * lg[3] + lg[4] * lg[4] + lg[5] * lg[5]);
}

// Degenerate geometry (the soft-failure branch the caller relies on is src/LidarOdometry.cpp:873-877).
// A solve is refused -- the iteration ends the align with SolverError, the pose stays the last one solved (the guess, if none) --
// when its system does not determine the pose, judged RELATIVELY (the CPU checker restates the same rule):
//   * Horn: fewer than 3 pairings; or the largest eigenvalue of N(S) is not separated from the second by more than kSingularRel of
//     the largest magnitude -- a line (S of rank 1: the rotation about it is free) -- or S itself is zero to rounding (all queries
//     in one point);
//   * Gauss-Newton / covariance: a pivot of the 6 x 6 normal matrix not above kSingularRel of its largest entry -- one plane (rank
//     3), one line of normals, ...  (rounds 1-3 compared pivots with 1e-300: a rank-deficient system has pivots of 1e-16 of the
//     scale, passed, and produced a step of 1e+16).
// Rounding leaves ~1e-16 of the scale in a pivot that should be zero and the two sides build their systems in different orders,
// so the decision is taken five orders of magnitude away from both.
constexpr double kSingularRel = 1e-11;

// Cyclic Jacobi on a symmetric 4x4; returns the unit eigenvector of the largest eigenvalue and (gap_rel, may be null) how far the
// largest eigenvalue is from the second, relative to the largest eigenvalue magnitude.
inline void max_eigvec_sym4(double A[4][4], double q[4], double* gap_rel = nullptr)
{
    double V[4][4] = {{1, 0, 0, 0}, {0, 1, 0, 0}, {0, 0, 1, 0}, {0, 0, 0, 1}};
    for (int sweep = 0; sweep < 50; ++sweep) {
        double off = 0, dg = 0;
        for (int i = 0; i < 4; ++i) {
            dg += A[i][i] * A[i][i];
            for (int j = i + 1; j < 4; ++j) off += A[i][j] * A[i][j];
        }
        if (off == 0 || off < 1e-34 * dg) break;
        for (int p = 0; p < 3; ++p)
            for (int r = p + 1; r < 4; ++r) {
                const double apr = A[p][r];
                if (apr == 0) continue;
                const double tau = (A[r][r] - A[p][p]) / (2 * apr);
                const double t = (tau >= 0 ? 1.0 : -1.0) / (std::fabs(tau) + std::sqrt(1 + tau * tau));
                const double c = 1 / std::sqrt(1 + t * t), s = t * c;
                for (int k = 0; k < 4; ++k) {  // A <- A J
                    const double x = A[k][p], y = A[k][r];
                    A[k][p] = c * x - s * y;
                    A[k][r] = s * x + c * y;
                }
                for (int k = 0; k < 4; ++k) {  // A <- J^T A
                    const double x = A[p][k], y = A[r][k];
                    A[p][k] = c * x - s * y;
                    A[r][k] = s * x + c * y;
                }
                for (int k = 0; k < 4; ++k) {
                    const double x = V[k][p], y = V[k][r];
                    V[k][p] = c * x - s * y;
                    V[k][r] = s * x + c * y;
                }
            }
    }
    int b = 0;
    for (int i = 1; i < 4; ++i)
        if (A[i][i] > A[b][b]) b = i;
    if (gap_rel) {
        double second = -1e300, mag = 0;
        for (int i = 0; i < 4; ++i) {
            if (i != b && A[i][i] > second) second = A[i][i];
            if (std::fabs(A[i][i]) > mag) mag = std::fabs(A[i][i]);
        }
        *gap_rel = mag > 0 ? (A[b][b] - second) / mag : 0.0;
    }
    double n = 0;
    for (int k = 0; k < 4; ++k) n += V[k][b] * V[k][b];
    n = std::sqrt(n);
    for (int k = 0; k < 4; ++k) q[k] = V[k][b] / n;
}

constexpr int kNAcc = 24;  // == MOLA_ICP_NACC

// Horn (1987) closed form from the accumulator block.  false if W <= 0 / degenerate.
inline bool solve_horn(const double acc[kNAcc], const double* cl_in, const double* cg_in, Mat4& T)
{
    const double W = acc[0];
    if (!(W > 0) || !std::isfinite(W)) return false;
    if (!(acc[16] >= 3.0)) return false;   // fewer than three pairings do not fix a rotation
    double cl[3], cg[3];
    for (int k = 0; k < 3; ++k) {
        cl[k] = cl_in ? cl_in[k] : acc[1 + k] / W;
        cg[k] = cg_in ? cg_in[k] : acc[4 + k] / W;
    }
    double S[3][3];
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c)
            S[r][c] = acc[7 + 3 * r + c] - cl[r] * acc[4 + c] - acc[1 + r] * cg[c] + W * cl[r] * cg[c];
    double N[4][4] = {
        {S[0][0] + S[1][1] + S[2][2], S[1][2] - S[2][1], S[2][0] - S[0][2], S[0][1] - S[1][0]},
        {S[1][2] - S[2][1], S[0][0] - S[1][1] - S[2][2], S[0][1] + S[1][0], S[2][0] + S[0][2]},
        {S[2][0] - S[0][2], S[0][1] + S[1][0], -S[0][0] + S[1][1] - S[2][2], S[1][2] + S[2][1]},
        {S[0][1] - S[1][0], S[2][0] + S[0][2], S[1][2] + S[2][1], -S[0][0] - S[1][1] + S[2][2]}};
    // (S is a difference of sums: when every query is the same point it is rounding noise around zero -- judged against the size
    //  of the sums it was formed from)
    double raw = 0, n_mag = 0;
    for (int k = 0; k < 9; ++k) raw = std::fmax(raw, std::fabs(acc[7 + k]));
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) raw = std::fmax(raw, std::fabs(W * cl[r] * cg[c]));
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) n_mag = std::fmax(n_mag, std::fabs(N[i][j]));
    if (!(n_mag > kSingularRel * raw)) return false;
    double q[4], gap = 0;
    max_eigvec_sym4(N, q, &gap);
    if (!(gap > kSingularRel)) return false;   // the rotation is not determined (kSingularRel, above)
    for (int k = 0; k < 4; ++k)
        if (!std::isfinite(q[k])) return false;
    double w = q[0], x = q[1], y = q[2], z = q[3];
    if (w < 0) { w = -w; x = -x; y = -y; z = -z; }
    T = Mat4::identity();
    T(0, 0) = 1 - 2 * (y * y + z * z); T(0, 1) = 2 * (x * y - w * z);     T(0, 2) = 2 * (x * z + w * y);
    T(1, 0) = 2 * (x * y + w * z);     T(1, 1) = 1 - 2 * (x * x + z * z); T(1, 2) = 2 * (y * z - w * x);
    T(2, 0) = 2 * (x * z - w * y);     T(2, 1) = 2 * (y * z + w * x);     T(2, 2) = 1 - 2 * (x * x + y * y);
    for (int r = 0; r < 3; ++r) T(r, 3) = cg[r] - (T(r, 0) * cl[0] + T(r, 1) * cl[1] + T(r, 2) * cl[2]);
    return true;
}

// 6x6 covariance of the pose (left perturbation, order x,y,z,wx,wy,wz):
// cov = sigma^2 (J^T J)^-1 with J_i = [I | -[p_i]x], p_i = T l_i, sigma^2 = sum d^2 / (3n-6).
// Own definition (the reference only consumes the mean: src/LidarOdometry.cpp:302,791).
inline bool pose_covariance(const double acc[kNAcc], const Mat4& T, double cov[36])
{
    std::memset(cov, 0, sizeof(double) * 36);
    const double W = acc[0], n = acc[16];
    if (!(W > 0) || n < 3) return false;
    double sl[3] = {acc[1], acc[2], acc[3]};
    double L[3][3] = {{acc[18], acc[19], acc[20]}, {acc[19], acc[21], acc[22]}, {acc[20], acc[22], acc[23]}};
    double R[3][3], t[3];
    for (int i = 0; i < 3; ++i) {
        for (int j = 0; j < 3; ++j) R[i][j] = T(i, j);
        t[i] = T(i, 3);
    }
    double sp[3], Rsl[3];
    for (int i = 0; i < 3; ++i) {
        Rsl[i] = R[i][0] * sl[0] + R[i][1] * sl[1] + R[i][2] * sl[2];
        sp[i] = Rsl[i] + W * t[i];
    }
    double RL[3][3], P[3][3];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) RL[i][j] = R[i][0] * L[0][j] + R[i][1] * L[1][j] + R[i][2] * L[2][j];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j)
            P[i][j] = RL[i][0] * R[j][0] + RL[i][1] * R[j][1] + RL[i][2] * R[j][2] + Rsl[i] * t[j] + t[i] * Rsl[j] +
                      W * t[i] * t[j];
    const double trP = P[0][0] + P[1][1] + P[2][2];
    double H[6][12] = {};
    for (int i = 0; i < 3; ++i) H[i][i] = W;
    // -[sp]x  in the upper-right block, its transpose (= [sp]x) lower-left
    const double sx[3][3] = {{0, -sp[2], sp[1]}, {sp[2], 0, -sp[0]}, {-sp[1], sp[0], 0}};
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            H[i][3 + j] = -sx[i][j];
            H[3 + i][j] = sx[i][j];
            H[3 + i][3 + j] = (i == j ? trP : 0.0) - P[i][j];
        }
    for (int i = 0; i < 6; ++i) H[i][6 + i] = 1.0;
    // Gauss-Jordan with partial pivoting (a pivot not above kSingularRel of the matrix' largest entry: singular)
    double h_scale = 0;
    for (int i = 0; i < 6; ++i)
        for (int j = 0; j < 6; ++j) h_scale = std::fmax(h_scale, std::fabs(H[i][j]));
    for (int c = 0; c < 6; ++c) {
        int piv = c;
        for (int r = c + 1; r < 6; ++r)
            if (std::fabs(H[r][c]) > std::fabs(H[piv][c])) piv = r;
        if (!(std::fabs(H[piv][c]) > kSingularRel * h_scale)) return false;
        if (piv != c)
            for (int k = 0; k < 12; ++k) { const double tmp = H[c][k]; H[c][k] = H[piv][k]; H[piv][k] = tmp; }
        const double inv = 1.0 / H[c][c];
        for (int k = 0; k < 12; ++k) H[c][k] *= inv;
        for (int r = 0; r < 6; ++r) {
            if (r == c) continue;
            const double f = H[r][c];
            if (f == 0) continue;
            for (int k = 0; k < 12; ++k) H[r][k] -= f * H[c][k];
        }
    }
    const double dof = 3.0 * n - 6.0;
    const double sigma2 = dof > 0 ? acc[17] / dof : 0.0;
    for (int i = 0; i < 6; ++i)
        for (int j = 0; j < 6; ++j) cov[6 * i + j] = sigma2 * H[i][6 + j];
    return true;
}

// SE(3) exponential of delta = (v, w): the 4x4 of exp([w]x, v)
inline Mat4 se3_exp(const double d[6])
{
    const double v[3] = {d[0], d[1], d[2]}, w[3] = {d[3], d[4], d[5]};
    const double th2 = w[0] * w[0] + w[1] * w[1] + w[2] * w[2], th = std::sqrt(th2);
    double a, b, c;  // R = I + a W + b W^2 ; V = I + b W + c W^2
    if (th < 1e-6) { a = 1 - th2 / 6; b = 0.5 - th2 / 24; c = 1.0 / 6 - th2 / 120; }
    else { a = std::sin(th) / th; b = (1 - std::cos(th)) / th2; c = (th - std::sin(th)) / (th2 * th); }
    const double W[3][3] = {{0, -w[2], w[1]}, {w[2], 0, -w[0]}, {-w[1], w[0], 0}};
    double W2[3][3];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            W2[i][j] = 0;
            for (int k = 0; k < 3; ++k) W2[i][j] += W[i][k] * W[k][j];
        }
    Mat4 T = Mat4::identity();
    for (int i = 0; i < 3; ++i) {
        double t = 0;
        for (int j = 0; j < 3; ++j) {
            T(i, j) = (i == j) + a * W[i][j] + b * W2[i][j];
            t += ((i == j) + b * W[i][j] + c * W2[i][j]) * v[j];
        }
        T(i, 3) = t;
    }
    return T;
}

constexpr int kNAccPlaneForm = 92;  // == kNAccPlaneHost: 78 (upper triangle, row-major a<=b) + 12 + 1 + count

// Gauss-Newton on the point-to-plane cost given as the quadratic form  f(x) = x^T A x - 2 b^T x + c0  in
// x = [R row-major, t]  (mp2p_icp::Solver_GaussNewton, params/icp-settings-regular.yaml:23-26: the same
// iterates as a Gauss-Newton over the pairings, left perturbation T <- exp(delta) T, stop at |delta| < 1e-7
// or after max_iters steps).  Returns false if fewer than 3 pairings or the normal equations are singular.
inline bool solve_gauss_newton_planes(const double acc[kNAccPlaneForm], const Mat4& T0, unsigned max_iters, Mat4& Tout,
                                      double* final_cost = nullptr, unsigned* iters_done = nullptr)
{
    if (!(acc[91] >= 3.0)) return false;
    double A[12][12], b[12];
    int q = 0;
    for (int i = 0; i < 12; ++i)
        for (int j = i; j < 12; ++j) { A[i][j] = acc[q]; A[j][i] = acc[q]; ++q; }
    for (int i = 0; i < 12; ++i) b[i] = acc[78 + i];
    const double c0 = acc[90];
    Mat4 T = T0;
    unsigned it = 0;
    double cost = 0;
    for (; it < max_iters; ++it) {
        double x[12];
        for (int r = 0; r < 3; ++r) {
            for (int c = 0; c < 3; ++c) x[3 * r + c] = T(r, c);
            x[9 + r] = T(r, 3);
        }
        // J = dx/d(delta), delta = (v, w), left perturbation: R' = R + [w]x R, t' = t + [w]x t + v
        double J[12][6] = {};
        for (int k = 0; k < 3; ++k) J[9 + k][k] = 1.0;
        for (int k = 0; k < 3; ++k) {
            double E[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};  // [e_k]x
            E[(k + 2) % 3][(k + 1) % 3] = 1.0;
            E[(k + 1) % 3][(k + 2) % 3] = -1.0;
            for (int r = 0; r < 3; ++r) {
                for (int c = 0; c < 3; ++c) {
                    double v = 0;
                    for (int m = 0; m < 3; ++m) v += E[r][m] * T(m, c);
                    J[3 * r + c][3 + k] = v;
                }
                double tv = 0;
                for (int m = 0; m < 3; ++m) tv += E[r][m] * T(m, 3);
                J[9 + r][3 + k] = tv;
            }
        }
        double Ax[12], res[12];
        for (int i = 0; i < 12; ++i) {
            double v = 0;
            for (int j = 0; j < 12; ++j) v += A[i][j] * x[j];
            Ax[i] = v;
            res[i] = v - b[i];  // (A x - b): gradient / 2 in x-space
        }
        cost = c0;
        for (int i = 0; i < 12; ++i) cost += x[i] * (Ax[i] - 2 * b[i]);
        double AJ[12][6];
        for (int i = 0; i < 12; ++i)
            for (int k = 0; k < 6; ++k) {
                double v = 0;
                for (int j = 0; j < 12; ++j) v += A[i][j] * J[j][k];
                AJ[i][k] = v;
            }
        double H[6][7];
        for (int a = 0; a < 6; ++a) {
            for (int k = 0; k < 6; ++k) {
                double v = 0;
                for (int i = 0; i < 12; ++i) v += J[i][a] * AJ[i][k];
                H[a][k] = v;
            }
            double g = 0;
            for (int i = 0; i < 12; ++i) g += J[i][a] * res[i];
            H[a][6] = -g;
        }
        // solve H d = -g (Gaussian elimination, partial pivoting; a pivot not above kSingularRel of the normal matrix' largest
        // entry: the pairings do not determine the pose -- one plane, one direction of normals -- SolverError)
        double h_scale = 0;
        for (int a = 0; a < 6; ++a)
            for (int k = 0; k < 6; ++k) h_scale = std::fmax(h_scale, std::fabs(H[a][k]));
        for (int c = 0; c < 6; ++c) {
            int piv = c;
            for (int r = c + 1; r < 6; ++r)
                if (std::fabs(H[r][c]) > std::fabs(H[piv][c])) piv = r;
            if (!(std::fabs(H[piv][c]) > kSingularRel * h_scale)) return false;
            if (piv != c)
                for (int k = 0; k < 7; ++k) { const double t = H[c][k]; H[c][k] = H[piv][k]; H[piv][k] = t; }
            for (int r = c + 1; r < 6; ++r) {
                const double f = H[r][c] / H[c][c];
                for (int k = c; k < 7; ++k) H[r][k] -= f * H[c][k];
            }
        }
        double d[6];
        for (int i = 5; i >= 0; --i) {
            double sacc = H[i][6];
            for (int j = i + 1; j < 6; ++j) sacc -= H[i][j] * d[j];
            d[i] = sacc / H[i][i];
        }
        for (int k = 0; k < 6; ++k)
            if (!std::isfinite(d[k])) return false;
        T = mul(se3_exp(d), T);
        double nd = 0;
        for (int k = 0; k < 6; ++k) nd += d[k] * d[k];
        if (std::sqrt(nd) < 1e-7) { ++it; break; }
    }
    Tout = T;
    if (final_cost) *final_cost = cost;
    if (iters_done) *iters_done = it;
    return true;
}

// 6x6 covariance of the pose for the point-to-plane cost (left perturbation, order x,y,z,wx,wy,wz -- as
// pose_covariance): cov = sigma^2 (J^T A J)^-1 with the quadratic form's A (the Gauss-Newton normal matrix of the
// last linearisation) at the pose T, sigma^2 = cost(T) / (n - 6).  Own definition: the reference only consumes the
// mean (src/LidarOdometry.cpp:302, 791).
inline bool pose_covariance_planes(const double acc[kNAccPlaneForm], const Mat4& T, double cov[36])
{
    std::memset(cov, 0, sizeof(double) * 36);
    const double n = acc[91];
    if (!(n > 6.0)) return false;
    double A[12][12], b[12], x[12];
    int q = 0;
    for (int i = 0; i < 12; ++i)
        for (int j = i; j < 12; ++j) { A[i][j] = acc[q]; A[j][i] = acc[q]; ++q; }
    for (int i = 0; i < 12; ++i) b[i] = acc[78 + i];
    for (int r = 0; r < 3; ++r) {
        for (int c = 0; c < 3; ++c) x[3 * r + c] = T(r, c);
        x[9 + r] = T(r, 3);
    }
    double J[12][6] = {};
    for (int k = 0; k < 3; ++k) J[9 + k][k] = 1.0;
    for (int k = 0; k < 3; ++k) {
        double E[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};  // [e_k]x
        E[(k + 2) % 3][(k + 1) % 3] = 1.0;
        E[(k + 1) % 3][(k + 2) % 3] = -1.0;
        for (int r = 0; r < 3; ++r) {
            for (int c = 0; c < 3; ++c) {
                double v = 0;
                for (int m = 0; m < 3; ++m) v += E[r][m] * T(m, c);
                J[3 * r + c][3 + k] = v;
            }
            double tv = 0;
            for (int m = 0; m < 3; ++m) tv += E[r][m] * T(m, 3);
            J[9 + r][3 + k] = tv;
        }
    }
    double cost = acc[90];
    for (int i = 0; i < 12; ++i) {
        double v = 0;
        for (int j = 0; j < 12; ++j) v += A[i][j] * x[j];
        cost += x[i] * (v - 2 * b[i]);
    }
    double H[6][12] = {};
    for (int a = 0; a < 6; ++a) {
        for (int k = 0; k < 6; ++k) {
            double v = 0;
            for (int i = 0; i < 12; ++i)
                for (int j = 0; j < 12; ++j) v += J[i][a] * A[i][j] * J[j][k];
            H[a][k] = v;
        }
        H[a][6 + a] = 1.0;
    }
    double h_scale = 0;
    for (int i = 0; i < 6; ++i)
        for (int j = 0; j < 6; ++j) h_scale = std::fmax(h_scale, std::fabs(H[i][j]));
    for (int c = 0; c < 6; ++c) {  // Gauss-Jordan with partial pivoting (singular: as above)
        int piv = c;
        for (int r = c + 1; r < 6; ++r)
            if (std::fabs(H[r][c]) > std::fabs(H[piv][c])) piv = r;
        if (!(std::fabs(H[piv][c]) > kSingularRel * h_scale)) return false;
        if (piv != c)
            for (int k = 0; k < 12; ++k) { const double t = H[c][k]; H[c][k] = H[piv][k]; H[piv][k] = t; }
        const double inv = 1.0 / H[c][c];
        for (int k = 0; k < 12; ++k) H[c][k] *= inv;
        for (int r = 0; r < 6; ++r) {
            if (r == c) continue;
            const double f = H[r][c];
            if (f == 0) continue;
            for (int k = 0; k < 12; ++k) H[r][k] -= f * H[c][k];
        }
    }
    const double sigma2 = (cost > 0 ? cost : 0.0) / (n - 6.0);
    for (int i = 0; i < 6; ++i)
        for (int j = 0; j < 6; ++j) {
            const double v = sigma2 * H[i][6 + j];
            if (!std::isfinite(v)) { std::memset(cov, 0, sizeof(double) * 36); return false; }
            cov[6 * i + j] = v;
        }
    return true;
}

}  // namespace mola_icp_amd
