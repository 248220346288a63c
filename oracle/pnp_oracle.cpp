// CPU oracle of the resection step: cv::solvePnPRansac(points_3d, pts, K, zeros,
// r, t, false, SOLVEPNP_EPNP) as called at /root/reference/bundle_adjustment.h:139-155.
// TEST INFRASTRUCTURE ONLY, like the rest of oracle/: nothing under sfm_amd/ may
// use it.
//
// The arithmetic lives in OpenCV 4.5.5 calib3d (solvepnp.cpp, epnp.cpp,
// ptsetreg.cpp), which is neither vendored in the reference nor installed here:
// everything below is restated from the published algorithm (Lepetit,
// Moreno-Noguer, Fua: "EPnP: an accurate O(n) solution to the PnP problem") and
// from recollection of that version's flow, and no fixture of the reference pins
// it -- PARITY UNPINNED.  What is pinned: known-answer properties (a planted
// pose is recovered, reprojection error ~ 0 on noise-free data) and bit equality
// with the GPU path on the same inputs.
//
// Restated flow [UPSTREAM, recalled]:
//   * object and image points are narrowed to float32 on entry;
//   * n < 4 is an error, n == 4 would use P3P (not on this path), n == 5 is one
//     direct EPnP solve with every point an inlier;
//   * otherwise RANSACPointSetRegistrator(model points 5, threshold
//     reprojectionError, confidence, iterationsCount): samples from cv::RNG(-1) by
//     getSubset (no subset check), one EPnP pose per sample, error = squared
//     reprojection distance evaluated in float32, inlier iff err <= (float)(thr*thr),
//     iteration count shrinking through RANSACUpdateNumIters;
//   * a final EPnP over all inliers gives the returned pose.
// Deviations where OpenCV's result is implementation defined: eigenvectors by
// cyclic Jacobi instead of cv::SVD (their signs cancel in the pose), small least
// squares systems by normal equations with pivoted elimination instead of SVD/QR
// solves, the rotation of the absolute-orientation step by the polar
// decomposition through a symmetric eigenproblem instead of a 3x3 SVD, and the
// rotation vector from the matrix without cv::Rodrigues' orthonormalising SVD.
// Two more places where this flow is known to differ from OpenCV's and a result
// can move in its last digits: hypotheses are scored with the rotation matrix the
// solver returned, where OpenCV projects with Rodrigues(Rodrigues(R)) of the
// stored rotation vector; and when the final pose over all inliers cannot be
// computed the best hypothesis is returned with success, where OpenCV returns
// false.
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

#include "sfm_oracle.h"

namespace {

// Symmetric eigenproblem by cyclic Jacobi rotations, row by row, at most 60
// sweeps; eigenvalues descending, eigenvector k in row k of V (like cv::SVD's
// U^T for a symmetric positive matrix).  A is destroyed.
template <int N>
void jacobi_eig(double* A, double* V, double* ev)
{
    for (int i = 0; i < N; i++)
        for (int j = 0; j < N; j++) V[i * N + j] = i == j ? 1.0 : 0.0;
    for (int sweep = 0; sweep < 60; sweep++) {
        double off = 0.0;
        for (int p = 0; p < N; p++)
            for (int q = p + 1; q < N; q++) off += A[p * N + q] * A[p * N + q];
        if (off == 0.0) break;
        for (int p = 0; p < N; p++)
            for (int q = p + 1; q < N; q++) {
                const double apq = A[p * N + q];
                if (apq == 0.0) continue;
                const double theta = (A[q * N + q] - A[p * N + p]) / (2.0 * apq);
                const double t = (theta >= 0.0 ? 1.0 : -1.0) / (std::fabs(theta) + std::sqrt(theta * theta + 1.0));
                const double c = 1.0 / std::sqrt(t * t + 1.0);
                const double s = t * c;
                for (int k = 0; k < N; k++) {  // columns p, q
                    const double akp = A[k * N + p], akq = A[k * N + q];
                    A[k * N + p] = c * akp - s * akq;
                    A[k * N + q] = s * akp + c * akq;
                }
                for (int k = 0; k < N; k++) {  // rows p, q
                    const double apk = A[p * N + k], aqk = A[q * N + k];
                    A[p * N + k] = c * apk - s * aqk;
                    A[q * N + k] = s * apk + c * aqk;
                }
                for (int k = 0; k < N; k++) {  // eigenvectors as rows of V
                    const double vpk = V[p * N + k], vqk = V[q * N + k];
                    V[p * N + k] = c * vpk - s * vqk;
                    V[q * N + k] = s * vpk + c * vqk;
                }
            }
    }
    for (int i = 0; i < N; i++) ev[i] = A[i * N + i];
    for (int i = 0; i < N; i++) {  // selection sort, descending, first maximum
        int b = i;
        for (int j = i + 1; j < N; j++)
            if (ev[j] > ev[b]) b = j;
        if (b != i) {
            const double t = ev[i]; ev[i] = ev[b]; ev[b] = t;
            for (int k = 0; k < N; k++) { const double w = V[i * N + k]; V[i * N + k] = V[b * N + k]; V[b * N + k] = w; }
        }
    }
}

// Least squares A x = b (rows x K, K <= 5) by normal equations and elimination
// with partial pivoting (first maximum).  False when a pivot is zero.
template <int K>
bool lsq(const double* A, const double* b, int rows, double* x)
{
    double G[K][K + 1];
    for (int i = 0; i < K; i++) {
        for (int j = 0; j < K; j++) {
            double s = 0.0;
            for (int r = 0; r < rows; r++) s += A[r * K + i] * A[r * K + j];
            G[i][j] = s;
        }
        double s = 0.0;
        for (int r = 0; r < rows; r++) s += A[r * K + i] * b[r];
        G[i][K] = s;
    }
    for (int p = 0; p < K; p++) {
        int bi = p;
        for (int i = p + 1; i < K; i++)
            if (std::fabs(G[i][p]) > std::fabs(G[bi][p])) bi = i;
        if (G[bi][p] == 0.0 || !(std::fabs(G[bi][p]) > 0.0)) return false;
        if (bi != p)
            for (int j = 0; j <= K; j++) { const double t = G[p][j]; G[p][j] = G[bi][j]; G[bi][j] = t; }
        for (int i = p + 1; i < K; i++) {
            const double f = G[i][p] / G[p][p];
            for (int j = p; j <= K; j++) G[i][j] = G[i][j] - f * G[p][j];
        }
    }
    for (int i = K - 1; i >= 0; i--) {
        double s = G[i][K];
        for (int j = i + 1; j < K; j++) s = s - G[i][j] * x[j];
        x[i] = s / G[i][i];
    }
    return true;
}

bool inv3(const double* m, double* o)
{
    const double c00 = m[4] * m[8] - m[5] * m[7], c01 = m[5] * m[6] - m[3] * m[8], c02 = m[3] * m[7] - m[4] * m[6];
    const double det = m[0] * c00 + m[1] * c01 + m[2] * c02;
    if (det == 0.0 || !(std::fabs(det) > 0.0)) return false;
    const double id = 1.0 / det;
    o[0] = c00 * id; o[1] = (m[2] * m[7] - m[1] * m[8]) * id; o[2] = (m[1] * m[5] - m[2] * m[4]) * id;
    o[3] = c01 * id; o[4] = (m[0] * m[8] - m[2] * m[6]) * id; o[5] = (m[2] * m[3] - m[0] * m[5]) * id;
    o[6] = c02 * id; o[7] = (m[1] * m[6] - m[0] * m[7]) * id; o[8] = (m[0] * m[4] - m[1] * m[3]) * id;
    return true;
}

struct Cam { double fu, fv, uc, vc; };

// EPnP on n points: pw[3n] world points, us[2n] pixel coordinates.  R (row
// major) and t of x_cam = R x_world + t.  False when a step degenerates.
bool epnp(const Cam& cam, const double* pw, const double* us, int n, double* R, double* t)
{
    // control points: centroid + principal directions
    double cws[4][3];
    for (int j = 0; j < 3; j++) {
        double s = 0.0;
        for (int i = 0; i < n; i++) s += pw[3 * i + j];
        cws[0][j] = s / n;
    }
    double C[9], Vc[9], dc[3];
    for (int a = 0; a < 3; a++)
        for (int b = 0; b < 3; b++) {
            double s = 0.0;
            for (int i = 0; i < n; i++) s += (pw[3 * i + a] - cws[0][a]) * (pw[3 * i + b] - cws[0][b]);
            C[3 * a + b] = s;
        }
    jacobi_eig<3>(C, Vc, dc);
    for (int i = 1; i < 4; i++) {
        const double k = std::sqrt(dc[i - 1] / n);
        for (int j = 0; j < 3; j++) cws[i][j] = cws[0][j] + k * Vc[3 * (i - 1) + j];
    }
    // barycentric coordinates
    double cc[9], ci[9];
    for (int i = 0; i < 3; i++)
        for (int j = 1; j < 4; j++) cc[3 * i + j - 1] = cws[j][i] - cws[0][i];
    if (!inv3(cc, ci)) return false;
    std::vector<double> alphas(4 * (size_t)n), pcs(3 * (size_t)n);
    for (int i = 0; i < n; i++) {
        const double* pi = pw + 3 * i;
        double* a = &alphas[4 * i];
        for (int j = 0; j < 3; j++)
            a[1 + j] = ci[3 * j] * (pi[0] - cws[0][0]) + ci[3 * j + 1] * (pi[1] - cws[0][1]) +
                       ci[3 * j + 2] * (pi[2] - cws[0][2]);
        a[0] = 1.0 - a[1] - a[2] - a[3];
    }
    // M^T M, entry by entry, rows of M in order (2 per point)
    auto mrow = [&](int r, int col) -> double {
        const int i = r >> 1, k = col / 3, c = col - 3 * k;
        const double as = alphas[4 * i + k];
        if ((r & 1) == 0) return c == 0 ? as * cam.fu : c == 1 ? 0.0 : as * (cam.uc - us[2 * i]);
        return c == 0 ? 0.0 : c == 1 ? as * cam.fv : as * (cam.vc - us[2 * i + 1]);
    };
    double MtM[144], Ut[144], ev[12];
    for (int a = 0; a < 12; a++)
        for (int b = a; b < 12; b++) {
            double s = 0.0;
            for (int r = 0; r < 2 * n; r++) s += mrow(r, a) * mrow(r, b);
            MtM[12 * a + b] = s;
            MtM[12 * b + a] = s;
        }
    jacobi_eig<12>(MtM, Ut, ev);
    // L (6 x 10) and rho
    const double* v[4] = {Ut + 12 * 11, Ut + 12 * 10, Ut + 12 * 9, Ut + 12 * 8};
    double dv[4][6][3];
    for (int i = 0; i < 4; i++) {
        int a = 0, b = 1;
        for (int j = 0; j < 6; j++) {
            for (int k = 0; k < 3; k++) dv[i][j][k] = v[i][3 * a + k] - v[i][3 * b + k];
            b++;
            if (b > 3) { a++; b = a + 1; }
        }
    }
    auto dot3 = [](const double* x, const double* y) { return x[0] * y[0] + x[1] * y[1] + x[2] * y[2]; };
    double L[60], rho[6];
    for (int i = 0; i < 6; i++) {
        double* row = L + 10 * i;
        row[0] = dot3(dv[0][i], dv[0][i]);
        row[1] = 2.0 * dot3(dv[0][i], dv[1][i]);
        row[2] = dot3(dv[1][i], dv[1][i]);
        row[3] = 2.0 * dot3(dv[0][i], dv[2][i]);
        row[4] = 2.0 * dot3(dv[1][i], dv[2][i]);
        row[5] = dot3(dv[2][i], dv[2][i]);
        row[6] = 2.0 * dot3(dv[0][i], dv[3][i]);
        row[7] = 2.0 * dot3(dv[1][i], dv[3][i]);
        row[8] = 2.0 * dot3(dv[2][i], dv[3][i]);
        row[9] = dot3(dv[3][i], dv[3][i]);
    }
    {
        int a = 0, b = 1;
        for (int j = 0; j < 6; j++) {
            const double d0 = cws[a][0] - cws[b][0], d1 = cws[a][1] - cws[b][1], d2 = cws[a][2] - cws[b][2];
            rho[j] = d0 * d0 + d1 * d1 + d2 * d2;
            b++;
            if (b > 3) { a++; b = a + 1; }
        }
    }
    // three initial guesses of the betas, each refined by five Gauss-Newton steps
    double betas[3][4];
    bool have[3] = {false, false, false};
    {
        double A4[24], b4[4];
        for (int i = 0; i < 6; i++) {
            A4[4 * i] = L[10 * i]; A4[4 * i + 1] = L[10 * i + 1]; A4[4 * i + 2] = L[10 * i + 3]; A4[4 * i + 3] = L[10 * i + 6];
        }
        if (lsq<4>(A4, rho, 6, b4)) {
            double* be = betas[0];
            if (b4[0] < 0) { be[0] = std::sqrt(-b4[0]); be[1] = -b4[1] / be[0]; be[2] = -b4[2] / be[0]; be[3] = -b4[3] / be[0]; }
            else { be[0] = std::sqrt(b4[0]); be[1] = b4[1] / be[0]; be[2] = b4[2] / be[0]; be[3] = b4[3] / be[0]; }
            have[0] = true;
        }
        double A3[18], b3[3];
        for (int i = 0; i < 6; i++) { A3[3 * i] = L[10 * i]; A3[3 * i + 1] = L[10 * i + 1]; A3[3 * i + 2] = L[10 * i + 2]; }
        if (lsq<3>(A3, rho, 6, b3)) {
            double* be = betas[1];
            if (b3[0] < 0) { be[0] = std::sqrt(-b3[0]); be[1] = b3[2] < 0 ? std::sqrt(-b3[2]) : 0.0; }
            else { be[0] = std::sqrt(b3[0]); be[1] = b3[2] > 0 ? std::sqrt(b3[2]) : 0.0; }
            if (b3[1] < 0) be[0] = -be[0];
            be[2] = 0.0;
            be[3] = 0.0;
            have[1] = true;
        }
        double A5[30], b5[5];
        for (int i = 0; i < 6; i++)
            for (int j = 0; j < 5; j++) A5[5 * i + j] = L[10 * i + j];
        if (lsq<5>(A5, rho, 6, b5)) {
            double* be = betas[2];
            if (b5[0] < 0) { be[0] = std::sqrt(-b5[0]); be[1] = b5[2] < 0 ? std::sqrt(-b5[2]) : 0.0; }
            else { be[0] = std::sqrt(b5[0]); be[1] = b5[2] > 0 ? std::sqrt(b5[2]) : 0.0; }
            if (b5[1] < 0) be[0] = -be[0];
            be[2] = b5[3] / be[0];
            be[3] = 0.0;
            have[2] = true;
        }
    }
    double best_err = 0.0;
    int best = -1;
    for (int c = 0; c < 3; c++) {
        if (!have[c]) continue;
        double* be = betas[c];
        bool ok = true;
        for (int it = 0; it < 5 && ok; it++) {
            double A[24], bb[6], x[4];
            for (int i = 0; i < 6; i++) {
                const double* rl = L + 10 * i;
                double* ra = A + 4 * i;
                ra[0] = 2 * rl[0] * be[0] + rl[1] * be[1] + rl[3] * be[2] + rl[6] * be[3];
                ra[1] = rl[1] * be[0] + 2 * rl[2] * be[1] + rl[4] * be[2] + rl[7] * be[3];
                ra[2] = rl[3] * be[0] + rl[4] * be[1] + 2 * rl[5] * be[2] + rl[8] * be[3];
                ra[3] = rl[6] * be[0] + rl[7] * be[1] + rl[8] * be[2] + 2 * rl[9] * be[3];
                bb[i] = rho[i] - (rl[0] * be[0] * be[0] + rl[1] * be[0] * be[1] + rl[2] * be[1] * be[1] +
                                  rl[3] * be[0] * be[2] + rl[4] * be[1] * be[2] + rl[5] * be[2] * be[2] +
                                  rl[6] * be[0] * be[3] + rl[7] * be[1] * be[3] + rl[8] * be[2] * be[3] +
                                  rl[9] * be[3] * be[3]);
            }
            ok = lsq<4>(A, bb, 6, x);
            if (ok)
                for (int i = 0; i < 4; i++) be[i] += x[i];
        }
        if (!ok) continue;
        // camera-frame control points and points
        double ccs[4][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
        for (int i = 0; i < 4; i++) {
            const double* vv = Ut + 12 * (11 - i);
            for (int j = 0; j < 4; j++)
                for (int k = 0; k < 3; k++) ccs[j][k] += be[i] * vv[3 * j + k];
        }
        for (int i = 0; i < n; i++) {
            const double* a = &alphas[4 * i];
            for (int j = 0; j < 3; j++)
                pcs[3 * i + j] = a[0] * ccs[0][j] + a[1] * ccs[1][j] + a[2] * ccs[2][j] + a[3] * ccs[3][j];
        }
        if (pcs[2] < 0.0)
            for (size_t i = 0; i < pcs.size(); i++) pcs[i] = -pcs[i];
        // absolute orientation
        double pc0[3], pw0[3];
        for (int j = 0; j < 3; j++) {
            double sc = 0.0, sw = 0.0;
            for (int i = 0; i < n; i++) { sc += pcs[3 * i + j]; sw += pw[3 * i + j]; }
            pc0[j] = sc / n;
            pw0[j] = sw / n;
        }
        double abt[9];
        for (int j = 0; j < 3; j++)
            for (int k = 0; k < 3; k++) {
                double s = 0.0;
                for (int i = 0; i < n; i++) s += (pcs[3 * i + j] - pc0[j]) * (pw[3 * i + k] - pw0[k]);
                abt[3 * j + k] = s;
            }
        // rotation of the polar decomposition: R = B (B^T B)^(-1/2)
        double S[9], Vs[9], es[3];
        for (int a = 0; a < 3; a++)
            for (int b = 0; b < 3; b++) S[3 * a + b] = abt[a] * abt[b] + abt[3 + a] * abt[3 + b] + abt[6 + a] * abt[6 + b];
        jacobi_eig<3>(S, Vs, es);
        if (!(es[2] > 0.0)) continue;  // rank-deficient configuration
        double W[9];  // (B^T B)^(-1/2) = sum_k v_k v_k^T / sqrt(e_k)
        for (int a = 0; a < 3; a++)
            for (int b = 0; b < 3; b++)
                W[3 * a + b] = Vs[a] * Vs[b] / std::sqrt(es[0]) + Vs[3 + a] * Vs[3 + b] / std::sqrt(es[1]) +
                               Vs[6 + a] * Vs[6 + b] / std::sqrt(es[2]);
        double Rc[9], tc[3];
        for (int a = 0; a < 3; a++)
            for (int b = 0; b < 3; b++) Rc[3 * a + b] = abt[3 * a] * W[b] + abt[3 * a + 1] * W[3 + b] + abt[3 * a + 2] * W[6 + b];
        const double det = Rc[0] * (Rc[4] * Rc[8] - Rc[5] * Rc[7]) - Rc[1] * (Rc[3] * Rc[8] - Rc[5] * Rc[6]) +
                           Rc[2] * (Rc[3] * Rc[7] - Rc[4] * Rc[6]);
        if (det < 0) { Rc[6] = -Rc[6]; Rc[7] = -Rc[7]; Rc[8] = -Rc[8]; }
        for (int a = 0; a < 3; a++) tc[a] = pc0[a] - (Rc[3 * a] * pw0[0] + Rc[3 * a + 1] * pw0[1] + Rc[3 * a + 2] * pw0[2]);
        // mean reprojection distance
        double sum = 0.0;
        for (int i = 0; i < n; i++) {
            const double* p = pw + 3 * i;
            const double Xc = Rc[0] * p[0] + Rc[1] * p[1] + Rc[2] * p[2] + tc[0];
            const double Yc = Rc[3] * p[0] + Rc[4] * p[1] + Rc[5] * p[2] + tc[1];
            const double iz = 1.0 / (Rc[6] * p[0] + Rc[7] * p[1] + Rc[8] * p[2] + tc[2]);
            const double du = us[2 * i] - (cam.uc + cam.fu * Xc * iz), dv2 = us[2 * i + 1] - (cam.vc + cam.fv * Yc * iz);
            sum += std::sqrt(du * du + dv2 * dv2);
        }
        const double err = sum / n;
        if (!(err == err)) continue;  // NaN
        if (best < 0 || err < best_err) {
            best = c;
            best_err = err;
            std::memcpy(R, Rc, sizeof(Rc));
            std::memcpy(t, tc, sizeof(tc));
        }
    }
    return best >= 0;
}

// Rotation vector of a rotation matrix (the inverse of cv::Rodrigues(rvec, R)).
void rotation_vector(const double* R, double* r)
{
    double rx = R[7] - R[5], ry = R[2] - R[6], rz = R[3] - R[1];
    const double s = std::sqrt((rx * rx + ry * ry + rz * rz) * 0.25);
    double c = (R[0] + R[4] + R[8] - 1.0) * 0.5;
    c = c > 1.0 ? 1.0 : c < -1.0 ? -1.0 : c;
    const double theta = std::acos(c);
    if (s < 1e-5) {
        if (c > 0) { r[0] = r[1] = r[2] = 0.0; return; }
        double t = (R[0] + 1.0) * 0.5;
        rx = std::sqrt(t > 0.0 ? t : 0.0);
        t = (R[4] + 1.0) * 0.5;
        ry = std::sqrt(t > 0.0 ? t : 0.0) * (R[1] < 0 ? -1.0 : 1.0);
        t = (R[8] + 1.0) * 0.5;
        rz = std::sqrt(t > 0.0 ? t : 0.0) * (R[2] < 0 ? -1.0 : 1.0);
        if (std::fabs(rx) < std::fabs(ry) && std::fabs(rx) < std::fabs(rz) && (R[5] > 0) != (ry * rz > 0)) rz = -rz;
        const double k = theta / std::sqrt(rx * rx + ry * ry + rz * rz);
        r[0] = rx * k; r[1] = ry * k; r[2] = rz * k;
        return;
    }
    const double vth = theta / (2.0 * s);
    r[0] = rx * vth; r[1] = ry * vth; r[2] = rz * vth;
}

}  // namespace

extern "C" {

int oracle_rotation_vector(const double* R, double* rvec)
{
    rotation_vector(R, rvec);
    return 1;
}

int oracle_epnp(const double* object_pts, const double* image_pts, int n, const double* K, double* R, double* t)
{
    if (n < 4) return -1;
    const Cam cam = {K[0], K[4], K[2], K[5]};
    return epnp(cam, object_pts, image_pts, n, R, t) ? 1 : 0;
}

int oracle_pnp_ransac(const double* object_pts, const double* image_pts, int n, const double* K,
                      int iterations_count, double reprojection_error, double confidence, double* rvec,
                      double* tvec, double* R_out, uint8_t* mask, int32_t* n_inliers, int32_t* iters_used)
{
    if (n_inliers) *n_inliers = 0;
    if (iters_used) *iters_used = 0;
    if (n < 5 || !(confidence > 0 && confidence < 1)) return -1;
    const Cam cam = {K[0], K[4], K[2], K[5]};
    const double ifx = 1.0 / cam.fu, ify = 1.0 / cam.fv;
    // float32 narrowing of both point sets; the solver's pixel coordinates go
    // through the normalised float32 value and back, as undistortPoints with no
    // distortion followed by the EPnP constructor does
    std::vector<float> of(3 * (size_t)n), uf(2 * (size_t)n);
    // ... while the final pose over all inliers comes from solvePnP on the float32
    // pixels widened to double, whose undistortPoints keeps that depth: ur
    std::vector<double> pw(3 * (size_t)n), us(2 * (size_t)n), ur(2 * (size_t)n);
    for (int i = 0; i < 3 * n; i++) { of[i] = (float)object_pts[i]; pw[i] = (double)of[i]; }
    for (int i = 0; i < n; i++) {
        uf[2 * i] = (float)image_pts[2 * i];
        uf[2 * i + 1] = (float)image_pts[2 * i + 1];
        const float xn = (float)(((double)uf[2 * i] - cam.uc) * ifx), yn = (float)(((double)uf[2 * i + 1] - cam.vc) * ify);
        us[2 * i] = (double)xn * cam.fu + cam.uc;
        us[2 * i + 1] = (double)yn * cam.fv + cam.vc;
        ur[2 * i] = (((double)uf[2 * i] - cam.uc) * ifx) * cam.fu + cam.uc;
        ur[2 * i + 1] = (((double)uf[2 * i + 1] - cam.vc) * ify) * cam.fv + cam.vc;
    }
    double R[9], t[3];
    auto finish = [&](const double* Rm, const double* tm) {
        rotation_vector(Rm, rvec);
        std::memcpy(tvec, tm, 3 * sizeof(double));
        if (R_out) std::memcpy(R_out, Rm, 9 * sizeof(double));
    };
    if (n == 5) {
        if (!epnp(cam, pw.data(), us.data(), 5, R, t)) return 0;
        finish(R, t);
        std::memset(mask, 1, 5);
        if (n_inliers) *n_inliers = 5;
        return 1;
    }
    const int max_it = iterations_count > 1 ? iterations_count : 1;
    std::vector<int32_t> table(5 * (size_t)max_it);
    if (oracle_sample_table(n, 5, max_it, table.data()) != 0) return 0;
    const float thr = (float)(reprojection_error * reprojection_error);
    int niters = max_it, best = 0, iter = 0;
    std::vector<uint8_t> cur(n);
    double bestR[9], bestT[3];
    for (; iter < niters; iter++) {
        double spw[15], sus[10];
        for (int k = 0; k < 5; k++) {
            const int id = table[5 * (size_t)iter + k];
            for (int j = 0; j < 3; j++) spw[3 * k + j] = pw[3 * id + j];
            sus[2 * k] = us[2 * id];
            sus[2 * k + 1] = us[2 * id + 1];
        }
        if (!epnp(cam, spw, sus, 5, R, t)) continue;
        int good = 0;
        for (int i = 0; i < n; i++) {
            const double* p = &pw[3 * i];
            const double X = R[0] * p[0] + R[1] * p[1] + R[2] * p[2] + t[0];
            const double Y = R[3] * p[0] + R[4] * p[1] + R[5] * p[2] + t[1];
            const double Z = R[6] * p[0] + R[7] * p[1] + R[8] * p[2] + t[2];
            const double iz = Z != 0.0 ? 1.0 / Z : 1.0;  // cv::projectPoints: z = z ? 1./z : 1
            const float pu = (float)(X * iz * cam.fu + cam.uc), pv = (float)(Y * iz * cam.fv + cam.vc);
            const float du = uf[2 * i] - pu, dv = uf[2 * i + 1] - pv;
            const float e = du * du + dv * dv;
            const int f = e <= thr;
            cur[i] = (uint8_t)f;
            good += f;
        }
        if (good > (best > 4 ? best : 4)) {
            best = good;
            std::memcpy(mask, cur.data(), n);
            std::memcpy(bestR, R, sizeof(R));
            std::memcpy(bestT, t, sizeof(t));
            niters = oracle_ransac_update_num_iters(confidence, (double)(n - good) / n, 5, niters);
        }
    }
    if (iters_used) *iters_used = iter;
    if (best <= 0) return 0;
    if (n_inliers) *n_inliers = best;
    // final pose from every inlier
    std::vector<double> ipw, ius;
    for (int i = 0; i < n; i++)
        if (mask[i]) {
            ipw.insert(ipw.end(), &pw[3 * i], &pw[3 * i] + 3);
            ius.push_back(ur[2 * i]);
            ius.push_back(ur[2 * i + 1]);
        }
    if (epnp(cam, ipw.data(), ius.data(), best, R, t)) finish(R, t);
    else finish(bestR, bestT);
    return 1;
}

}  // extern "C"
