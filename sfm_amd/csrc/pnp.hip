// Resection: the work of
//   solvePnPRansac(points_3d, pts, cameraMatrix, distortion_coeffs, r, t, false, SOLVEPNP_EPNP)
// at /root/reference/bundle_adjustment.h:139-155 (Resection(), called from
// /root/reference/main.cpp:520), i.e. OpenCV's RANSAC registrator around the
// EPnP solver with model size 5, followed by one EPnP over all inliers.
//
// One wavefront solves one EPnP problem (a five-point hypothesis, or the final
// pose over thousands of inliers).  Every sum over the points is owned by one
// lane, entry by entry, and runs over the points in order, so each entry sees
// exactly the operation order of a sequential loop; the small dense algebra
// (Jacobi eigenproblems of size 3 and 12, the least squares steps for the betas)
// runs on the first lanes.  Unfused double arithmetic (-ffp-contract=off).
// Scoring: one thread per (hypothesis, point), reprojection distance in float32
// like cv::projectPoints into a CV_32F array followed by a float32 norm.
// The sample stream (cv::RNG(-1), getSubset) and the order-dependent iteration
// count (RANSACUpdateNumIters) are integer / scalar host logic.
#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstring>
#include <vector>

#include "sfm_ctx.h"

namespace sfm {

namespace {

struct PnpCam {
    double fu, fv, uc, vc;
};

// ---- pieces of the solver that run on a few lanes, data in LDS ----------------

// Symmetric eigenproblem, cyclic Jacobi, row by row, at most 60 sweeps; lanes
// 0..N-1 apply each rotation (lane k updates element k of the two columns, rows
// and eigenvector rows), every lane evaluates the rotation itself.  Eigenvalues
// descending in ev, eigenvector k in row k of V.  Called by the whole wavefront.
template <int N>
__device__ void jacobi_eig_wave(double* A, double* V, double* ev, int lane)
{
    if (lane < N)
        for (int j = 0; j < N; j++) V[lane * N + j] = lane == j ? 1.0 : 0.0;
    __syncthreads();
    for (int sweep = 0; sweep < 60; sweep++) {
        double off = 0.0;
        for (int p = 0; p < N; p++)
            for (int q = p + 1; q < N; q++) off += A[p * N + q] * A[p * N + q];
        if (off == 0.0) break;  // the same value on every lane
        for (int p = 0; p < N; p++)
            for (int q = p + 1; q < N; q++) {
                const double apq = A[p * N + q];
                if (apq == 0.0) continue;
                const double theta = (A[q * N + q] - A[p * N + p]) / (2.0 * apq);
                const double t = (theta >= 0.0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
                const double c = 1.0 / sqrt(t * t + 1.0);
                const double s = t * c;
                __syncthreads();  // everybody has read the rotation's inputs
                if (lane < N) {
                    const int k = lane;
                    const double akp = A[k * N + p], akq = A[k * N + q];
                    A[k * N + p] = c * akp - s * akq;
                    A[k * N + q] = s * akp + c * akq;
                }
                __syncthreads();
                if (lane < N) {
                    const int k = lane;
                    const double apk = A[p * N + k], aqk = A[q * N + k];
                    A[p * N + k] = c * apk - s * aqk;
                    A[q * N + k] = s * apk + c * aqk;
                    const double vpk = V[p * N + k], vqk = V[q * N + k];
                    V[p * N + k] = c * vpk - s * vqk;
                    V[q * N + k] = s * vpk + c * vqk;
                }
                __syncthreads();
            }
    }
    __syncthreads();
    if (lane == 0) {
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
    __syncthreads();
}

// Least squares A x = b (rows x K) by normal equations and elimination with
// partial pivoting (first maximum); one lane.
template <int K>
__device__ bool lsq(const double* A, const double* b, int rows, double* x)
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
            if (fabs(G[i][p]) > fabs(G[bi][p])) bi = i;
        if (G[bi][p] == 0.0 || !(fabs(G[bi][p]) > 0.0)) return false;
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

__device__ bool inv3(const double* m, double* o)
{
    const double c00 = m[4] * m[8] - m[5] * m[7], c01 = m[5] * m[6] - m[3] * m[8], c02 = m[3] * m[7] - m[4] * m[6];
    const double det = m[0] * c00 + m[1] * c01 + m[2] * c02;
    if (det == 0.0 || !(fabs(det) > 0.0)) return false;
    const double id = 1.0 / det;
    o[0] = c00 * id; o[1] = (m[2] * m[7] - m[1] * m[8]) * id; o[2] = (m[1] * m[5] - m[2] * m[4]) * id;
    o[3] = c01 * id; o[4] = (m[0] * m[8] - m[2] * m[6]) * id; o[5] = (m[2] * m[3] - m[0] * m[5]) * id;
    o[6] = c02 * id; o[7] = (m[1] * m[6] - m[0] * m[7]) * id; o[8] = (m[0] * m[4] - m[1] * m[3]) * id;
    return true;
}

// LDS state of one EPnP problem
struct EpnpShared {
    double cws[4][3];
    double C[9], Vc[9], dc[3];
    double ci[9];
    double MtM[144], Ut[144], ev[12];
    double L[60], rho[6];
    double betas[3][4];
    double ccs[3][4][3];
    double pc0[3], pw0[3], abt[9], S[9], Vs[9], es[3];
    double R[9], t[3];
    double best_err;
    int have[3];
    int ok, best, sign_neg;
};

// EPnP over n points: pw[3n] object points, us[2n] pixel coordinates (global
// memory), work = 7n doubles of scratch (alphas 4n, camera-frame points 3n).
// Result in sh.R / sh.t, sh.ok.  One wavefront, every lane calls.
__device__ void epnp_wave(const PnpCam cam, const double* __restrict__ pw, const double* __restrict__ us, int n,
                          double* __restrict__ work, EpnpShared& sh, int lane)
{
    double* alphas = work;
    double* pcs = work + 4 * (size_t)n;
    if (lane == 0) { sh.ok = 1; sh.best = -1; }
    // control points: centroid + principal directions
    if (lane < 3) {
        double s = 0.0;
        for (int i = 0; i < n; i++) s += pw[3 * i + lane];
        sh.cws[0][lane] = s / n;
    }
    __syncthreads();
    if (lane < 9) {
        const int a = lane / 3, b = lane - 3 * a;
        const double ca = sh.cws[0][a], cb = sh.cws[0][b];
        double s = 0.0;
        for (int i = 0; i < n; i++) s += (pw[3 * i + a] - ca) * (pw[3 * i + b] - cb);
        sh.C[lane] = s;
    }
    __syncthreads();
    jacobi_eig_wave<3>(sh.C, sh.Vc, sh.dc, lane);
    if (lane == 0) {
        for (int i = 1; i < 4; i++) {
            const double k = sqrt(sh.dc[i - 1] / n);
            for (int j = 0; j < 3; j++) sh.cws[i][j] = sh.cws[0][j] + k * sh.Vc[3 * (i - 1) + j];
        }
        double cc[9];
        for (int i = 0; i < 3; i++)
            for (int j = 1; j < 4; j++) cc[3 * i + j - 1] = sh.cws[j][i] - sh.cws[0][i];
        if (!inv3(cc, sh.ci)) sh.ok = 0;
    }
    __syncthreads();
    if (!sh.ok) return;
    // barycentric coordinates
    for (int i = lane; i < n; i += 64) {
        const double* pi = pw + 3 * i;
        double* a = alphas + 4 * i;
        double a1[3];
        for (int j = 0; j < 3; j++)
            a1[j] = sh.ci[3 * j] * (pi[0] - sh.cws[0][0]) + sh.ci[3 * j + 1] * (pi[1] - sh.cws[0][1]) +
                    sh.ci[3 * j + 2] * (pi[2] - sh.cws[0][2]);
        a[1] = a1[0];
        a[2] = a1[1];
        a[3] = a1[2];
        a[0] = 1.0 - a1[0] - a1[1] - a1[2];
    }
    __syncthreads();
    // M^T M: 78 entries of the upper triangle, entry per lane, rows of M in order
    for (int e = lane; e < 78; e += 64) {
        int a = 0, rem = e;
        while (rem >= 12 - a) { rem -= 12 - a; a++; }
        const int b = a + rem;
        const int ka = a / 3, ca = a - 3 * ka, kb = b / 3, cb = b - 3 * kb;
        double s = 0.0;
        for (int i = 0; i < n; i++) {
            const double asa = alphas[4 * i + ka], asb = alphas[4 * i + kb];
            const double u = us[2 * i], v = us[2 * i + 1];
            // even row of the point, then its odd row
            const double ea = ca == 0 ? asa * cam.fu : ca == 1 ? 0.0 : asa * (cam.uc - u);
            const double eb = cb == 0 ? asb * cam.fu : cb == 1 ? 0.0 : asb * (cam.uc - u);
            s += ea * eb;
            const double oa = ca == 0 ? 0.0 : ca == 1 ? asa * cam.fv : asa * (cam.vc - v);
            const double ob = cb == 0 ? 0.0 : cb == 1 ? asb * cam.fv : asb * (cam.vc - v);
            s += oa * ob;
        }
        sh.MtM[12 * a + b] = s;
        sh.MtM[12 * b + a] = s;
    }
    __syncthreads();
    jacobi_eig_wave<12>(sh.MtM, sh.Ut, sh.ev, lane);
    if (lane == 0) {
        const double* v[4] = {sh.Ut + 12 * 11, sh.Ut + 12 * 10, sh.Ut + 12 * 9, sh.Ut + 12 * 8};
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
        double* L = sh.L;
        double* rho = sh.rho;
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
                const double d0 = sh.cws[a][0] - sh.cws[b][0], d1 = sh.cws[a][1] - sh.cws[b][1],
                             d2 = sh.cws[a][2] - sh.cws[b][2];
                rho[j] = d0 * d0 + d1 * d1 + d2 * d2;
                b++;
                if (b > 3) { a++; b = a + 1; }
            }
        }
        // three initial guesses of the betas
        sh.have[0] = sh.have[1] = sh.have[2] = 0;
        {
            double A4[24], b4[4];
            for (int i = 0; i < 6; i++) {
                A4[4 * i] = L[10 * i]; A4[4 * i + 1] = L[10 * i + 1]; A4[4 * i + 2] = L[10 * i + 3]; A4[4 * i + 3] = L[10 * i + 6];
            }
            if (lsq<4>(A4, rho, 6, b4)) {
                double* be = sh.betas[0];
                if (b4[0] < 0) { be[0] = sqrt(-b4[0]); be[1] = -b4[1] / be[0]; be[2] = -b4[2] / be[0]; be[3] = -b4[3] / be[0]; }
                else { be[0] = sqrt(b4[0]); be[1] = b4[1] / be[0]; be[2] = b4[2] / be[0]; be[3] = b4[3] / be[0]; }
                sh.have[0] = 1;
            }
            double A3[18], b3[3];
            for (int i = 0; i < 6; i++) { A3[3 * i] = L[10 * i]; A3[3 * i + 1] = L[10 * i + 1]; A3[3 * i + 2] = L[10 * i + 2]; }
            if (lsq<3>(A3, rho, 6, b3)) {
                double* be = sh.betas[1];
                if (b3[0] < 0) { be[0] = sqrt(-b3[0]); be[1] = b3[2] < 0 ? sqrt(-b3[2]) : 0.0; }
                else { be[0] = sqrt(b3[0]); be[1] = b3[2] > 0 ? sqrt(b3[2]) : 0.0; }
                if (b3[1] < 0) be[0] = -be[0];
                be[2] = 0.0;
                be[3] = 0.0;
                sh.have[1] = 1;
            }
            double A5[30], b5[5];
            for (int i = 0; i < 6; i++)
                for (int j = 0; j < 5; j++) A5[5 * i + j] = L[10 * i + j];
            if (lsq<5>(A5, rho, 6, b5)) {
                double* be = sh.betas[2];
                if (b5[0] < 0) { be[0] = sqrt(-b5[0]); be[1] = b5[2] < 0 ? sqrt(-b5[2]) : 0.0; }
                else { be[0] = sqrt(b5[0]); be[1] = b5[2] > 0 ? sqrt(b5[2]) : 0.0; }
                if (b5[1] < 0) be[0] = -be[0];
                be[2] = b5[3] / be[0];
                be[3] = 0.0;
                sh.have[2] = 1;
            }
        }
        // five Gauss-Newton steps each, then the camera-frame control points
        for (int c = 0; c < 3; c++) {
            if (!sh.have[c]) continue;
            double* be = sh.betas[c];
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
            if (!ok) { sh.have[c] = 0; continue; }
            for (int j = 0; j < 4; j++)
                for (int k = 0; k < 3; k++) sh.ccs[c][j][k] = 0.0;
            for (int i = 0; i < 4; i++) {
                const double* vv = sh.Ut + 12 * (11 - i);
                for (int j = 0; j < 4; j++)
                    for (int k = 0; k < 3; k++) sh.ccs[c][j][k] += be[i] * vv[3 * j + k];
            }
        }
    }
    __syncthreads();
    // absolute orientation and reprojection error of each candidate
    for (int c = 0; c < 3; c++) {
        if (!sh.have[c]) continue;  // the same on every lane
        for (int i = lane; i < n; i += 64) {
            const double* a = alphas + 4 * i;
            for (int j = 0; j < 3; j++)
                pcs[3 * i + j] = a[0] * sh.ccs[c][0][j] + a[1] * sh.ccs[c][1][j] + a[2] * sh.ccs[c][2][j] +
                                 a[3] * sh.ccs[c][3][j];
        }
        __syncthreads();
        if (lane == 0) sh.sign_neg = pcs[2] < 0.0;
        __syncthreads();
        const double sg = sh.sign_neg ? -1.0 : 1.0;  // the flip of every camera-frame point is an exact negation
        if (lane < 3) {
            double s = 0.0;
            for (int i = 0; i < n; i++) s += sg * pcs[3 * i + lane];
            sh.pc0[lane] = s / n;
        } else if (lane < 6) {
            const int j = lane - 3;
            double s = 0.0;
            for (int i = 0; i < n; i++) s += pw[3 * i + j];
            sh.pw0[j] = s / n;
        }
        __syncthreads();
        if (lane < 9) {
            const int j = lane / 3, k = lane - 3 * j;
            const double cj = sh.pc0[j], wk = sh.pw0[k];
            double s = 0.0;
            for (int i = 0; i < n; i++) s += (sg * pcs[3 * i + j] - cj) * (pw[3 * i + k] - wk);
            sh.abt[lane] = s;
        }
        __syncthreads();
        if (lane < 9) {
            const int a = lane / 3, b = lane - 3 * a;
            sh.S[lane] = sh.abt[a] * sh.abt[b] + sh.abt[3 + a] * sh.abt[3 + b] + sh.abt[6 + a] * sh.abt[6 + b];
        }
        __syncthreads();
        jacobi_eig_wave<3>(sh.S, sh.Vs, sh.es, lane);
        if (lane == 0 && sh.es[2] > 0.0) {
            const double* Vs = sh.Vs;
            const double* es = sh.es;
            const double* abt = sh.abt;
            double W[9], Rc[9], tc[3];
            for (int a = 0; a < 3; a++)
                for (int b = 0; b < 3; b++)
                    W[3 * a + b] = Vs[a] * Vs[b] / sqrt(es[0]) + Vs[3 + a] * Vs[3 + b] / sqrt(es[1]) +
                                   Vs[6 + a] * Vs[6 + b] / sqrt(es[2]);
            for (int a = 0; a < 3; a++)
                for (int b = 0; b < 3; b++)
                    Rc[3 * a + b] = abt[3 * a] * W[b] + abt[3 * a + 1] * W[3 + b] + abt[3 * a + 2] * W[6 + b];
            const double det = Rc[0] * (Rc[4] * Rc[8] - Rc[5] * Rc[7]) - Rc[1] * (Rc[3] * Rc[8] - Rc[5] * Rc[6]) +
                               Rc[2] * (Rc[3] * Rc[7] - Rc[4] * Rc[6]);
            if (det < 0) { Rc[6] = -Rc[6]; Rc[7] = -Rc[7]; Rc[8] = -Rc[8]; }
            for (int a = 0; a < 3; a++)
                tc[a] = sh.pc0[a] - (Rc[3 * a] * sh.pw0[0] + Rc[3 * a + 1] * sh.pw0[1] + Rc[3 * a + 2] * sh.pw0[2]);
            double sum = 0.0;
            for (int i = 0; i < n; i++) {
                const double* p = pw + 3 * i;
                const double Xc = Rc[0] * p[0] + Rc[1] * p[1] + Rc[2] * p[2] + tc[0];
                const double Yc = Rc[3] * p[0] + Rc[4] * p[1] + Rc[5] * p[2] + tc[1];
                const double iz = 1.0 / (Rc[6] * p[0] + Rc[7] * p[1] + Rc[8] * p[2] + tc[2]);
                const double du = us[2 * i] - (cam.uc + cam.fu * Xc * iz), dv2 = us[2 * i + 1] - (cam.vc + cam.fv * Yc * iz);
                sum += sqrt(du * du + dv2 * dv2);
            }
            const double err = sum / n;
            if (err == err && (sh.best < 0 || err < sh.best_err)) {
                sh.best = c;
                sh.best_err = err;
                for (int a = 0; a < 9; a++) sh.R[a] = Rc[a];
                for (int a = 0; a < 3; a++) sh.t[a] = tc[a];
            }
        }
        __syncthreads();
    }
    if (lane == 0 && sh.best < 0) sh.ok = 0;
    __syncthreads();
}

// ---- kernels -------------------------------------------------------------------

// Narrowing of both point sets to float32 and the solver's pixel coordinates, as
// the OpenCV front end does: the hypotheses see them through the normalised
// float32 value and back (us), the final pose over all inliers through a
// normalisation kept in double (ur: solvePnP is handed the float32 pixels widened
// to double there, and undistortPoints keeps its input's depth).
__global__ void __launch_bounds__(256) pnp_prep_kernel(const double* __restrict__ obj, const double* __restrict__ img,
                                                       int n, PnpCam cam, double* __restrict__ pw,
                                                       double* __restrict__ us, float* __restrict__ uf,
                                                       double* __restrict__ ur)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const double ifx = 1.0 / cam.fu, ify = 1.0 / cam.fv;
    for (int j = 0; j < 3; j++) pw[3 * i + j] = (double)(float)obj[3 * i + j];
    const float u = (float)img[2 * i], v = (float)img[2 * i + 1];
    uf[2 * i] = u;
    uf[2 * i + 1] = v;
    const float xn = (float)(((double)u - cam.uc) * ifx), yn = (float)(((double)v - cam.vc) * ify);
    us[2 * i] = (double)xn * cam.fu + cam.uc;
    us[2 * i + 1] = (double)yn * cam.fv + cam.vc;
    ur[2 * i] = (((double)u - cam.uc) * ifx) * cam.fu + cam.uc;
    ur[2 * i + 1] = (((double)v - cam.vc) * ify) * cam.fv + cam.vc;
}

// One workgroup of one wavefront per problem.  table != null: problem h is the
// five points table[5h..5h+4] of (pw, us); otherwise the single problem is the
// first *n_ptr points.  pose[h] = {R[9], t[3], ok}.
__global__ void __launch_bounds__(64) pnp_solve_kernel(const double* __restrict__ pw, const double* __restrict__ us,
                                                       const int32_t* __restrict__ table,
                                                       const int32_t* __restrict__ n_ptr, PnpCam cam,
                                                       double* __restrict__ work, size_t work_stride,
                                                       double* __restrict__ pose)
{
    __shared__ EpnpShared sh;
    __shared__ double s_pw[15], s_us[10];
    const int lane = threadIdx.x, h = blockIdx.x;
    double* wk = work + (size_t)h * work_stride;
    double* out = pose + 13 * (size_t)h;
    if (table) {
        // the sample's points, gathered next to the problem's scratch
        double* gpw = wk, *gus = wk + 15;
        if (lane < 5) {
            const int id = table[5 * h + lane];
            for (int j = 0; j < 3; j++) gpw[3 * lane + j] = pw[3 * id + j];
            gus[2 * lane] = us[2 * id];
            gus[2 * lane + 1] = us[2 * id + 1];
        }
        __syncthreads();
        epnp_wave(cam, gpw, gus, 5, wk + 32, sh, lane);
    } else {
        const int n = *n_ptr;
        if (n < 4) {
            if (lane == 0) out[12] = 0.0;
            return;
        }
        epnp_wave(cam, pw, us, n, wk, sh, lane);
    }
    if (lane < 9) out[lane] = sh.R[lane];
    if (lane < 3) out[9 + lane] = sh.t[lane];
    if (lane == 0) out[12] = sh.ok ? 1.0 : 0.0;
    (void)s_pw;
    (void)s_us;
}

// counts[h] += inliers of pose h among the n points: cv::projectPoints in double,
// the projected pixel stored as float32, squared distance in float32.
__global__ void __launch_bounds__(256) pnp_score_kernel(const double* __restrict__ pw, const float* __restrict__ uf,
                                                        int n, PnpCam cam, const double* __restrict__ pose, float thr,
                                                        int32_t* __restrict__ counts, uint8_t* __restrict__ mask_out,
                                                        int only_hyp)
{
    const int h = only_hyp >= 0 ? only_hyp : blockIdx.y;
    const double* P = pose + 13 * (size_t)h;
    const int i = blockIdx.x * 256 + threadIdx.x;
    int f = 0;
    if (P[12] != 0.0 && i < n) {
        const double* p = pw + 3 * i;
        const double X = P[0] * p[0] + P[1] * p[1] + P[2] * p[2] + P[9];
        const double Y = P[3] * p[0] + P[4] * p[1] + P[5] * p[2] + P[10];
        const double Z = P[6] * p[0] + P[7] * p[1] + P[8] * p[2] + P[11];
        const double iz = Z != 0.0 ? 1.0 / Z : 1.0;
        const float pu = (float)(X * iz * cam.fu + cam.uc), pv = (float)(Y * iz * cam.fv + cam.vc);
        const float du = uf[2 * i] - pu, dv = uf[2 * i + 1] - pv;
        const float e = du * du + dv * dv;
        f = e <= thr;
    }
    if (mask_out) {
        if (i < n) mask_out[i] = (uint8_t)f;
        return;
    }
    const unsigned long long b = __ballot(f);
    if ((threadIdx.x & 63) == 0 && b) atomicAdd(&counts[h], __popcll(b));
}

// Inlier points in order, packed for the final solve: one workgroup.
__global__ void __launch_bounds__(1024) pnp_compact_kernel(const uint8_t* __restrict__ mask, int n,
                                                           const double* __restrict__ pw, const double* __restrict__ us,
                                                           double* __restrict__ ipw, double* __restrict__ ius,
                                                           int32_t* __restrict__ n_out)
{
    __shared__ int s_wave[16];
    __shared__ int s_base;
    if (threadIdx.x == 0) s_base = 0;
    __syncthreads();
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int i0 = 0; i0 < n; i0 += 1024) {
        const int i = i0 + threadIdx.x;
        const int f = i < n && mask[i];
        const unsigned long long b = __ballot(f);
        if (lane == 0) s_wave[wave] = __popcll(b);
        __syncthreads();
        int off = s_base;
        for (int w = 0; w < wave; w++) off += s_wave[w];
        if (f) {
            const int o = off + __popcll(b & ((1ull << lane) - 1ull));
            for (int j = 0; j < 3; j++) ipw[3 * o + j] = pw[3 * i + j];
            ius[2 * o] = us[2 * i];
            ius[2 * o + 1] = us[2 * i + 1];
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            int tot = 0;
            for (int w = 0; w < 16; w++) tot += s_wave[w];
            s_base += tot;
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) *n_out = s_base;
}

int cv_round(double v) { return (int)std::nearbyint(v); }

// cv::RANSACUpdateNumIters (OpenCV 4.5.5 calib3d ptsetreg.cpp).
int update_num_iters(double p, double ep, int model_points, int max_iters)
{
    p = std::max(p, 0.);
    p = std::min(p, 1.);
    ep = std::max(ep, 0.);
    ep = std::min(ep, 1.);
    double num = std::max(1. - p, DBL_MIN);
    double denom = 1. - std::pow(1. - ep, model_points);
    if (denom < DBL_MIN) return 0;
    num = std::log(num);
    denom = std::log(denom);
    return denom >= 0 || -num >= max_iters * (-denom) ? max_iters : cv_round(num / denom);
}

// cv::RNG (multiply-with-carry) and RANSACPointSetRegistrator::getSubset for a
// model whose subsets need no check: rows of `m` distinct indices below `count`.
void sample_rows(int count, int m, int rows, std::vector<int32_t>& out)
{
    uint64_t state = 0xffffffffffffffffULL;
    out.resize((size_t)rows * m);
    for (int r = 0; r < rows; r++) {
        int32_t* idx = &out[(size_t)r * m];
        for (int i = 0; i < m;) {
            state = (uint64_t)(uint32_t)state * 4164903690U + (uint32_t)(state >> 32);
            const int v = (int)((uint32_t)state % (uint32_t)count);
            int j = 0;
            for (; j < i; j++)
                if (idx[j] == v) break;
            if (j < i) continue;
            idx[i++] = v;
        }
    }
}

// Rotation vector of a rotation matrix (what cv::Rodrigues(rvec, R) inverts).
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

size_t align256(size_t v) { return (v + 255) / 256 * 256; }

}  // namespace
}  // namespace sfm

using namespace sfm;

extern "C" int sfm_gpu_pnp_ransac(sfm_gpu_ctx* ctx, const double* object_pts, const double* image_pts, int n,
                                  const double* K, int iterations_count, double reprojection_error,
                                  double confidence, double* rvec_out, double* tvec_out, double* R_out,
                                  uint8_t* inlier_mask, int32_t* n_inliers, int32_t* iters_used, int32_t* ok)
{
    if (!ctx) return SFM_GPU_ERR_ARG;
    if (!ok || !rvec_out || !tvec_out || !K || n < 0 || (n > 0 && (!object_pts || !image_pts)))
        return ctx->fail(SFM_GPU_ERR_ARG, "null pointer or negative count");
    *ok = 0;
    if (n_inliers) *n_inliers = 0;
    if (iters_used) *iters_used = 0;
    if (n < 4) return ctx->fail(SFM_GPU_ERR_ARG, "solvePnPRansac needs at least 4 correspondences, got %d", n);
    if (n == 4)
        return ctx->fail(SFM_GPU_ERR_UNSUPPORTED, "4 correspondences: OpenCV switches to P3P, which is not on this path");
    if (!(confidence > 0 && confidence < 1)) return ctx->fail(SFM_GPU_ERR_ARG, "confidence must be in (0,1)");
    if (!(reprojection_error > 0) || !std::isfinite(reprojection_error))
        return ctx->fail(SFM_GPU_ERR_ARG, "reprojection_error must be positive and finite");
    if (!std::isfinite(K[0]) || !std::isfinite(K[4]) || K[0] == 0.0 || K[4] == 0.0 || !std::isfinite(K[2]) ||
        !std::isfinite(K[5]))
        return ctx->fail(SFM_GPU_ERR_ARG, "camera matrix needs finite, non-zero focal lengths");
    const int max_it = std::max(iterations_count, 1);
    if (max_it > SFM_GPU_MAX_HYPOTHESES)
        return ctx->fail(SFM_GPU_ERR_CAPACITY, "iterations_count %d exceeds SFM_GPU_MAX_HYPOTHESES", iterations_count);
    SFM_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t s = ctx->stream;
    const PnpCam cam = {K[0], K[4], K[2], K[5]};
    const int n_hyp = n == 5 ? 0 : max_it;

    // device layout: inputs | pw | us | uf | inlier pw | inlier us | mask | table | poses | counts | n_inl | work
    const size_t in_b = align256((size_t)n * 40);
    const size_t pw_b = align256((size_t)n * 24), us_b = align256((size_t)n * 16), uf_b = align256((size_t)n * 8);
    const size_t mask_b = align256((size_t)n), table_b = align256((size_t)std::max(n_hyp, 1) * 20);
    const size_t pose_b = align256((size_t)(n_hyp + 1) * 13 * 8), cnt_b = align256((size_t)(n_hyp + 1) * 4 + 16);
    const size_t hyp_work = 32 + 7 * 5 + 5;  // doubles per hypothesis: gathered sample + solver scratch
    const size_t work_b = align256(std::max((size_t)n_hyp * hyp_work, (size_t)7 * n + 64) * 8);
    const size_t total = in_b + 2 * (pw_b + us_b) + us_b + uf_b + mask_b + table_b + pose_b + cnt_b + work_b;
    SFM_HIP(ctx, ctx->pnp_ws.reserve(total));
    char* base = ctx->pnp_ws.as<char>();
    double* d_obj = (double*)base;
    double* d_img = d_obj + 3 * (size_t)n;
    double* d_pw = (double*)(base + in_b);
    double* d_us = (double*)((char*)d_pw + pw_b);
    float* d_uf = (float*)((char*)d_us + us_b);
    double* d_ipw = (double*)((char*)d_uf + uf_b);
    double* d_ius = (double*)((char*)d_ipw + pw_b);
    double* d_ur = (double*)((char*)d_ius + us_b);
    uint8_t* d_mask = (uint8_t*)((char*)d_ur + us_b);
    int32_t* d_table = (int32_t*)((char*)d_mask + mask_b);
    double* d_pose = (double*)((char*)d_table + table_b);
    int32_t* d_counts = (int32_t*)((char*)d_pose + pose_b);
    int32_t* d_ninl = d_counts + n_hyp + 1;
    double* d_work = (double*)((char*)d_counts + cnt_b);

    SFM_HIP(ctx, hipMemcpyAsync(d_obj, object_pts, (size_t)n * 24, hipMemcpyHostToDevice, s));
    SFM_HIP(ctx, hipMemcpyAsync(d_img, image_pts, (size_t)n * 16, hipMemcpyHostToDevice, s));
    {
        ProfScope ps(ctx, "pnp_prep");
        hipLaunchKernelGGL(pnp_prep_kernel, dim3((n + 255) / 256), dim3(256), 0, s, d_obj, d_img, n, cam, d_pw, d_us,
                           d_uf, d_ur);
        SFM_HIP(ctx, hipGetLastError());
    }
    const float thr = (float)(reprojection_error * reprojection_error);
    double pose[13];
    auto finish = [&](const double* P) {
        rotation_vector(P, rvec_out);
        std::memcpy(tvec_out, P + 9, 24);
        if (R_out) std::memcpy(R_out, P, 72);
    };
    if (n == 5) {
        // no RANSAC: one direct solve, every point an inlier
        SFM_HIP(ctx, hipMemcpyAsync(d_ninl, &n, 4, hipMemcpyHostToDevice, s));
        ProfScope ps(ctx, "pnp_solve");
        hipLaunchKernelGGL(pnp_solve_kernel, dim3(1), dim3(64), 0, s, d_pw, d_us, (const int32_t*)nullptr, d_ninl, cam,
                           d_work, (size_t)0, d_pose);
        SFM_HIP(ctx, hipGetLastError());
        SFM_HIP(ctx, hipMemcpyAsync(pose, d_pose, sizeof(pose), hipMemcpyDeviceToHost, s));
        SFM_HIP(ctx, hipStreamSynchronize(s));
        if (pose[12] == 0.0) return SFM_GPU_OK;
        finish(pose);
        if (inlier_mask) std::memset(inlier_mask, 1, 5);
        if (n_inliers) *n_inliers = 5;
        *ok = 1;
        return SFM_GPU_OK;
    }
    // every sample row of the stream up front (pure integer work), all hypotheses in one launch
    std::vector<int32_t> table;
    sample_rows(n, 5, n_hyp, table);
    SFM_HIP(ctx, hipMemcpyAsync(d_table, table.data(), table.size() * 4, hipMemcpyHostToDevice, s));
    SFM_HIP(ctx, hipMemsetAsync(d_counts, 0, (size_t)(n_hyp + 1) * 4, s));
    {
        ProfScope ps(ctx, "pnp_solve");
        hipLaunchKernelGGL(pnp_solve_kernel, dim3(n_hyp), dim3(64), 0, s, d_pw, d_us, d_table, (const int32_t*)nullptr,
                           cam, d_work, hyp_work, d_pose);
        SFM_HIP(ctx, hipGetLastError());
    }
    {
        ProfScope ps(ctx, "pnp_score");
        hipLaunchKernelGGL(pnp_score_kernel, dim3((n + 255) / 256, n_hyp), dim3(256), 0, s, d_pw, d_uf, n, cam, d_pose,
                           thr, d_counts, (uint8_t*)nullptr, -1);
        SFM_HIP(ctx, hipGetLastError());
    }
    std::vector<int32_t> counts(n_hyp);
    std::vector<double> poses((size_t)n_hyp * 13);
    SFM_HIP(ctx, hipMemcpyAsync(counts.data(), d_counts, (size_t)n_hyp * 4, hipMemcpyDeviceToHost, s));
    SFM_HIP(ctx, hipMemcpyAsync(poses.data(), d_pose, poses.size() * 8, hipMemcpyDeviceToHost, s));
    SFM_HIP(ctx, hipStreamSynchronize(s));
    // OpenCV's sequential loop over the counts: the iteration count shrinks
    // whenever a better pose turns up, hypotheses beyond it are never looked at
    int niters = max_it, best = 0, win = -1, iter = 0;
    for (; iter < niters; iter++) {
        if (poses[(size_t)iter * 13 + 12] == 0.0) continue;
        const int good = counts[iter];
        if (good > std::max(best, 4)) {
            best = good;
            win = iter;
            niters = update_num_iters(confidence, (double)(n - good) / n, 5, niters);
        }
    }
    if (iters_used) *iters_used = iter;
    if (win < 0) return SFM_GPU_OK;
    if (n_inliers) *n_inliers = best;
    // mask of the winner, then the final pose from every inlier
    {
        ProfScope ps(ctx, "pnp_score");
        hipLaunchKernelGGL(pnp_score_kernel, dim3((n + 255) / 256, 1), dim3(256), 0, s, d_pw, d_uf, n, cam, d_pose, thr,
                           d_counts, d_mask, win);
        hipLaunchKernelGGL(pnp_compact_kernel, dim3(1), dim3(1024), 0, s, d_mask, n, d_pw, d_ur, d_ipw, d_ius, d_ninl);
        SFM_HIP(ctx, hipGetLastError());
    }
    {
        ProfScope ps(ctx, "pnp_solve");
        hipLaunchKernelGGL(pnp_solve_kernel, dim3(1), dim3(64), 0, s, d_ipw, d_ius, (const int32_t*)nullptr, d_ninl, cam,
                           d_work, (size_t)0, d_pose + 13 * (size_t)n_hyp);
        SFM_HIP(ctx, hipGetLastError());
    }
    SFM_HIP(ctx, hipMemcpyAsync(pose, d_pose + 13 * (size_t)n_hyp, sizeof(pose), hipMemcpyDeviceToHost, s));
    if (inlier_mask) SFM_HIP(ctx, hipMemcpyAsync(inlier_mask, d_mask, (size_t)n, hipMemcpyDeviceToHost, s));
    SFM_HIP(ctx, hipStreamSynchronize(s));
    if (pose[12] != 0.0) finish(pose);
    else finish(&poses[(size_t)win * 13]);  // the refit degenerated: the hypothesis' own pose
    *ok = 1;
    return SFM_GPU_OK;
}
