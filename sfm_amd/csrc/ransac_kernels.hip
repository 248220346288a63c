// Batched RANSAC for gfx950: sample-table generation, one wavefront per
// hypothesis minimal solvers (five-point essential, seven-point fundamental),
// scoring of every candidate model against every correspondence, winner
// selection and inlier mask.
//
// Replaces the arithmetic behind
//     findEssentialMat(ptsL, ptsR, K, FM_RANSAC, 0.99, 1.0, inliers)
// at /root/reference/main.cpp:224 (OpenCV's RANSACPointSetRegistrator with
// the five-point callback), and its findFundamentalMat counterpart.
//
// Every floating-point result that decides an inlier is produced by the same
// sequence of IEEE double operations a sequential CPU implementation would
// use: lanes split the work by OUTPUT element (each element's own operation
// order is untouched), and this file is compiled with -ffp-contract=off so no
// multiply-add is fused.
#include <cfloat>
#include <climits>

#include "sfm_ctx.h"
#include "ransac_dev.h"

namespace sfm {

namespace {

// ---- monomial pair tables --------------------------------------------------
// Polynomials in (x,y,z):  degree<=1 [x,y,z,1];  degree<=2
// [x2,xy,xz,y2,yz,z2,x,y,z,1];  degree<=3 in Nister's elimination order
// [x3,y3,x2y,xy2,x2z,x2,y2z,y2,xyz,xy, xz2,xz,x,yz2,yz,y,z3,z2,z,1].
// P11[m] lists the (i,j) with mono1[i]*mono1[j] == mono2[m] in row-major
// order, P21[m] the (i,j) with mono2[i]*mono1[j] == mono3[m].
__device__ __constant__ const signed char P11N[10] = {1, 2, 2, 1, 2, 1, 2, 2, 2, 1};
__device__ __constant__ const signed char P11[10][2][2] = {
    {{0,0},{-1,-1}}, {{0,1},{1,0}}, {{0,2},{2,0}}, {{1,1},{-1,-1}}, {{1,2},{2,1}},
    {{2,2},{-1,-1}}, {{0,3},{3,0}}, {{1,3},{3,1}}, {{2,3},{3,2}}, {{3,3},{-1,-1}}};
__device__ __constant__ const signed char P21N[20] = {1, 1, 2, 2, 2, 2, 2, 2, 3, 3,
                                                      2, 3, 2, 2, 3, 2, 1, 2, 2, 1};
__device__ __constant__ const signed char P21[20][3][2] = {
    {{0,0},{-1,-1},{-1,-1}}, {{3,1},{-1,-1},{-1,-1}}, {{0,1},{1,0},{-1,-1}}, {{1,1},{3,0},{-1,-1}},
    {{0,2},{2,0},{-1,-1}},   {{0,3},{6,0},{-1,-1}},   {{3,2},{4,1},{-1,-1}}, {{3,3},{7,1},{-1,-1}},
    {{1,2},{2,1},{4,0}},     {{1,3},{6,1},{7,0}},     {{2,2},{5,0},{-1,-1}}, {{2,3},{6,2},{8,0}},
    {{6,3},{9,0},{-1,-1}},   {{4,2},{5,1},{-1,-1}},   {{4,3},{7,2},{8,1}},   {{7,3},{9,1},{-1,-1}},
    {{5,2},{-1,-1},{-1,-1}}, {{5,3},{8,2},{-1,-1}},   {{8,3},{9,2},{-1,-1}}, {{9,3},{-1,-1},{-1,-1}}};

// coefficient m of a*b, both degree<=1 in (x,y,z)
__device__ __forceinline__ double mul11_coef(const double* a, const double* b, int m)
{
    double s = 0.0;
    const int n = P11N[m];
    for (int t = 0; t < n; ++t) s = s + a[P11[m][t][0]] * b[P11[m][t][1]];
    return s;
}
// coefficient m of a*b, a degree<=2, b degree<=1
__device__ __forceinline__ double mul21_coef(const double* a, const double* b, int m)
{
    double s = 0.0;
    const int n = P21N[m];
    for (int t = 0; t < n; ++t) s = s + a[P21[m][t][0]] * b[P21[m][t][1]];
    return s;
}
// coefficient i of a*b, univariate, degrees na and nb
__device__ __forceinline__ double pmul_coef(const double* a, int na, const double* b, int nb, int i)
{
    double s = 0.0;
    for (int ia = 0; ia <= na; ++ia) {
        const int ib = i - ia;
        if (ib >= 0 && ib <= nb) s = s + a[ia] * b[ib];
    }
    return s;
}

__device__ __forceinline__ double horner(const double* c, int n, double x)
{
    double s = c[n];
    for (int i = n - 1; i >= 0; --i) s = s * x + c[i];
    return s;
}

// Brackets the root of a first / last interval of a level (which reaches out to
// the root bound) by doubling steps away from its finite end, where P is
// monotone; see the CPU statement of the same search in oracle/.
template <class HornerF>
__device__ __forceinline__ void tighten_open_interval(HornerF&& P, bool open_lo, bool open_hi, double& lo,
                                                      double& hi, double& flo, double& fhi)
{
    if (flo == 0.0 || fhi == 0.0 || ((flo < 0.0) == (fhi < 0.0))) return;
    if (open_hi) {
        double w = 0.0625 * fmax(1.0, fabs(lo));
        const double e = lo;
        double a = lo, fa = flo;
        for (;;) {
            const double x = e + w;
            if (!(x < hi)) break;
            const double fx = P(x);
            if (fx == 0.0 || ((fx < 0.0) != (flo < 0.0))) { hi = x; fhi = fx; break; }
            a = x;
            fa = fx;
            w = w * 2.0;
        }
        lo = a;
        flo = fa;
    } else if (open_lo) {
        double w = 0.0625 * fmax(1.0, fabs(hi));
        const double e = hi;
        double b = hi, fb = fhi;
        for (;;) {
            const double x = e - w;
            if (!(x > lo)) break;
            const double fx = P(x);
            if (fx == 0.0 || ((fx < 0.0) != (fhi < 0.0))) { lo = x; flo = fx; break; }
            b = x;
            fb = fx;
            w = w * 2.0;
        }
        hi = b;
        fhi = fb;
    }
}

// Per-wavefront workspace in LDS.
struct SolveWs {
    double Q[7 * 9];
    double N[4 * 9];
    double L[9][10];
    double T[3][10];
    double A[10][20];
    double B[3][3][5];
    double M[3][8];
    double c[11];
    int perm[9];
};

// Null space of the rows x 9 system in w.Q by Gauss-Jordan with complete
// pivoting (first maximum in row-major scan).  Writes 9-rows basis vectors
// into w.N.  One wavefront; returns false on a zero pivot.
__device__ bool null_space_9(SolveWs& w, int rows, int lane)
{
    double* A = w.Q;
    if (lane < 9) w.perm[lane] = lane;
    __syncthreads();
    for (int p = 0; p < rows; ++p) {
        // pivot = first maximum of |A[i][j]|, i >= p, j >= p, in row-major scan:
        // one element per lane, then a butterfly on (larger value, smaller index)
        double best = -1.0;
        int bidx = 0x7FFFFFFF;
        if (lane < rows * 9) {
            const int i = lane / 9, j = lane % 9;
            if (i >= p && j >= p) {
                best = fabs(A[lane]);
                bidx = lane;
            }
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const double ov = __shfl_xor(best, off);
            const int oi = __shfl_xor(bidx, off);
            if (ov > best || (ov == best && oi < bidx)) { best = ov; bidx = oi; }
        }
        if (!(best > 0.0)) return false;
        const int bi = bidx / 9, bj = bidx % 9;
        __syncthreads();
        if (bi != p && lane < 9) {
            const double t = A[p * 9 + lane];
            A[p * 9 + lane] = A[bi * 9 + lane];
            A[bi * 9 + lane] = t;
        }
        __syncthreads();
        if (bj != p) {
            if (lane < rows) {
                const double t = A[lane * 9 + p];
                A[lane * 9 + p] = A[lane * 9 + bj];
                A[lane * 9 + bj] = t;
            }
            if (lane == 63) {
                const int t = w.perm[p];
                w.perm[p] = w.perm[bj];
                w.perm[bj] = t;
            }
        }
        __syncthreads();
        const double piv = A[p * 9 + p];
        __syncthreads();
        if (lane > p && lane < 9) A[p * 9 + lane] = A[p * 9 + lane] / piv;
        __syncthreads();
        if (lane < rows * 9) {
            const int i = lane / 9, j = lane % 9;
            if (i != p && j > p) A[i * 9 + j] = A[i * 9 + j] - A[i * 9 + p] * A[p * 9 + j];
        }
        __syncthreads();
        if (lane < rows) A[lane * 9 + p] = (lane == p) ? 1.0 : 0.0;
        __syncthreads();
    }
    const int nfree = 9 - rows;
    if (lane < nfree * 9) {
        const int k = lane / 9, jj = lane % 9;
        double v;
        if (jj < rows) v = -A[jj * 9 + rows + k];
        else v = (jj == rows + k) ? 1.0 : 0.0;
        w.N[k * 9 + w.perm[jj]] = v;
    }
    __syncthreads();
    return true;
}

// Layout of what the first half of the five-point solver leaves in a
// hypothesis' slot of the models buffer (90 doubles) for the second half:
// the degree-10 polynomial, the rows of B(z) and the null-space basis.
constexpr int FP_C = 0;    // c[0..10]
constexpr int FP_B = 11;   // 3 rows x (4 + 4 + 5) coefficients
constexpr int FP_N = 50;   // N[4][9]

// Five-point essential matrix solver (Nister 2004), first half, one wavefront.
// x1s/x2s: the 5 normalised correspondences (uniform across the wave).
// Leaves polynomial, B(z) and null space in `slot`; false when the
// elimination meets a zero pivot (no models).
__device__ bool five_point_wave(SolveWs& w, const double* x1s, const double* x2s, double* slot, int lane)
{
    if (lane < 45) {
        const int i = lane / 9, j = lane % 9;
        const double u1 = x1s[2 * i], v1 = x1s[2 * i + 1], u2 = x2s[2 * i], v2 = x2s[2 * i + 1];
        double q;
        switch (j) {
            case 0: q = u2 * u1; break;
            case 1: q = u2 * v1; break;
            case 2: q = u2; break;
            case 3: q = v2 * u1; break;
            case 4: q = v2 * v1; break;
            case 5: q = v2; break;
            case 6: q = u1; break;
            case 7: q = v1; break;
            default: q = 1.0; break;
        }
        w.Q[lane] = q;
    }
    __syncthreads();
    if (!null_space_9(w, 5, lane)) return false;

    // entry (r,c) of E = xX + yY + zZ + W as a degree-1 polynomial: gathered
    // per use from N (e[r][c][k] = N[k*9 + 3r + c]).
    auto load_e = [&](int r, int c, double* e) {
#pragma unroll
        for (int k = 0; k < 4; ++k) e[k] = w.N[k * 9 + r * 3 + c];
    };
    // Lambda = E E^T - 0.5 tr(E E^T) I, upper triangle first
    if (lane < 60) {
        const int pair = lane / 10, m = lane % 10;
        const int pi[6] = {0, 0, 0, 1, 1, 2}, pj[6] = {0, 1, 2, 1, 2, 2};
        const int i = pi[pair], j = pj[pair];
        double a[4], b[4];
        load_e(i, 0, a); load_e(j, 0, b);
        const double t0 = mul11_coef(a, b, m);
        load_e(i, 1, a); load_e(j, 1, b);
        const double t1 = mul11_coef(a, b, m);
        load_e(i, 2, a); load_e(j, 2, b);
        const double t2 = mul11_coef(a, b, m);
        w.L[i * 3 + j][m] = (t0 + t1) + t2;
    }
    __syncthreads();
    if (lane < 10) {
        const int m = lane;
        const double tr = (w.L[0][m] + w.L[4][m]) + w.L[8][m];
        const double h = 0.5 * tr;
        w.L[0][m] = w.L[0][m] - h;
        w.L[4][m] = w.L[4][m] - h;
        w.L[8][m] = w.L[8][m] - h;
        w.L[3][m] = w.L[1][m];
        w.L[6][m] = w.L[2][m];
        w.L[7][m] = w.L[5][m];
    }
    __syncthreads();
    for (int o = lane; o < 180; o += 64) {
        const int row = o / 20, m = o % 20, i = row / 3, j = row % 3;
        double b[4];
        load_e(0, j, b);
        const double t0 = mul21_coef(w.L[i * 3 + 0], b, m);
        load_e(1, j, b);
        const double t1 = mul21_coef(w.L[i * 3 + 1], b, m);
        load_e(2, j, b);
        const double t2 = mul21_coef(w.L[i * 3 + 2], b, m);
        w.A[row][m] = (t0 + t1) + t2;
    }
    // determinant row: minors of rows 1,2 then expansion along row 0
    if (lane < 30) {
        const int which = lane / 10, m = lane % 10;
        const int ca[3] = {1, 0, 0}, cb[3] = {2, 2, 1};
        double a[4], b[4];
        load_e(1, ca[which], a); load_e(2, cb[which], b);
        const double p = mul11_coef(a, b, m);
        load_e(1, cb[which], a); load_e(2, ca[which], b);
        const double q = mul11_coef(a, b, m);
        w.T[which][m] = p - q;
    }
    __syncthreads();
    if (lane < 20) {
        const int m = lane;
        double b[4];
        load_e(0, 0, b);
        const double t0 = mul21_coef(w.T[0], b, m);
        load_e(0, 1, b);
        const double t1 = mul21_coef(w.T[1], b, m);
        load_e(0, 2, b);
        const double t2 = mul21_coef(w.T[2], b, m);
        w.A[9][m] = (t0 - t1) + t2;
    }
    __syncthreads();

    // Gauss-Jordan on the 10 leading columns, partial pivoting (first maximum),
    // in registers: lane j < 20 holds column j.  The pivot search is lane p's own
    // business, everything the other lanes need from it (pivot row number, pivot,
    // the multipliers A[i][p]) travels by readlane; no LDS, no barrier.  Same
    // operations on the same values as the row-by-row statement of the oracle.
    {
        double a[10];
#pragma unroll
        for (int i = 0; i < 10; ++i) a[i] = lane < 20 ? w.A[i][lane] : 0.0;
        auto bcast = [](double v, int src) -> double {
            const int hi = __builtin_amdgcn_readlane(__double2hiint(v), src);
            const int lo = __builtin_amdgcn_readlane(__double2loint(v), src);
            return __hiloint2double(hi, lo);
        };
        bool singular = false;
#pragma unroll
        for (int p = 0; p < 10; ++p) {
            int bi = p;
            double best = -1.0;
#pragma unroll
            for (int i = p; i < 10; ++i) {
                const double v = fabs(a[i]);
                if (v > best) { best = v; bi = i; }
            }
            bi = __builtin_amdgcn_readlane(bi, p);
            best = bcast(best, p);
            if (!(best > 0.0)) { singular = true; break; }
#pragma unroll
            for (int i = p + 1; i < 10; ++i)
                if (bi == i) {  // uniform
                    const double t = a[p];
                    a[p] = a[i];
                    a[i] = t;
                }
            const double piv = bcast(a[p], p);
            if (lane > p) a[p] = a[p] / piv;
#pragma unroll
            for (int i = 0; i < 10; ++i) {
                if (i == p) continue;
                const double f = bcast(a[i], p);
                if (lane > p) a[i] = a[i] - f * a[p];
            }
            if (lane == p) {
#pragma unroll
                for (int i = 0; i < 10; ++i) a[i] = (i == p) ? 1.0 : 0.0;
            }
        }
        if (singular) return false;
        // rows 4..9 of the trailing columns are all that is read from here on
        if (lane >= 10 && lane < 20) {
#pragma unroll
            for (int i = 4; i < 10; ++i) w.A[i][lane] = a[i];
        }
        __syncthreads();
    }

    // B(z): rows (e - z f), (g - z h), (i - z j) of the reduced system
    if (lane < 9) {
        const int r = lane / 3, v = lane % 3;
        const double* hi = w.A[4 + 2 * r];
        const double* lo = w.A[5 + 2 * r];
        double* b = w.B[r][v];
        if (v < 2) {
            const int c0 = 10 + 3 * v;
            b[0] = hi[c0 + 2];
            b[1] = hi[c0 + 1] - lo[c0 + 2];
            b[2] = hi[c0 + 0] - lo[c0 + 1];
            b[3] = -lo[c0 + 0];
            b[4] = 0.0;
        } else {
            b[0] = hi[19];
            b[1] = hi[18] - lo[19];
            b[2] = hi[17] - lo[18];
            b[3] = hi[16] - lo[17];
            b[4] = -lo[16];
        }
    }
    __syncthreads();
    // 2x2 minors of rows 1,2 (degrees 7,7,6), then det as degree-10 polynomial
    if (lane < 24) {
        const int which = lane / 8, i = lane % 8;
        double v = 0.0;
        if (which == 0) v = pmul_coef(w.B[1][1], 3, w.B[2][2], 4, i) - pmul_coef(w.B[1][2], 4, w.B[2][1], 3, i);
        else if (which == 1) v = pmul_coef(w.B[1][0], 3, w.B[2][2], 4, i) - pmul_coef(w.B[1][2], 4, w.B[2][0], 3, i);
        else if (i < 7) v = pmul_coef(w.B[1][0], 3, w.B[2][1], 3, i) - pmul_coef(w.B[1][1], 3, w.B[2][0], 3, i);
        w.M[which][i] = v;
    }
    __syncthreads();
    if (lane < 11) {
        const int i = lane;
        const double t0 = pmul_coef(w.B[0][0], 3, w.M[0], 7, i);
        const double t1 = pmul_coef(w.B[0][1], 3, w.M[1], 7, i);
        const double t2 = pmul_coef(w.B[0][2], 4, w.M[2], 6, i);
        w.c[i] = (t0 - t1) + t2;
    }
    __syncthreads();
    if (lane < 11) slot[FP_C + lane] = w.c[lane];
    if (lane < 39) {
        const int r = lane / 13, t = lane % 13;
        slot[FP_B + lane] = t < 4 ? w.B[r][0][t] : (t < 8 ? w.B[r][1][t - 4] : w.B[r][2][t - 8]);
    }
    if (lane < 36) slot[FP_N + lane] = w.N[lane];
    return true;
}



// ---- seven-point solver, one THREAD per hypothesis -----------------------------
// The seven-point algorithm is short and almost entirely sequential, so 64
// hypotheses share a wavefront, each lane running the sequential statement of
// the algorithm.  The 7x9 system lives in LDS as element-major columns
// (element e of lane l at sA[e*64 + l]): any per-lane choice of pivot row or
// column then hits the lane's own bank pair, so the dynamically indexed swaps
// of complete pivoting are conflict-free.

template <int NA, int NB>
__device__ __forceinline__ void pmul_t(const double* a, const double* b, double* out)
{
#pragma unroll
    for (int i = 0; i <= NA + NB; ++i) out[i] = 0.0;
#pragma unroll
    for (int i = 0; i <= NA; ++i)
#pragma unroll
        for (int j = 0; j <= NB; ++j) out[i + j] = out[i + j] + a[i] * b[j];
}

template <int N>
__device__ __forceinline__ double horner_t(const double* c, double x)
{
    double s = c[N];
#pragma unroll
    for (int i = N - 1; i >= 0; --i) s = s * x + c[i];
    return s;
}

// Newton with bisection safeguard run to a fixed point (see the CPU statement of
// the same iteration in oracle/ for the reasoning); the degree is a compile-time
// constant so that the coefficient arrays stay in registers.
template <int N>
__device__ __forceinline__ double solve_bracket_t(const double* P, const double* dP, double lo, double hi,
                                                  bool lo_neg)
{
    double xl = lo_neg ? lo : hi;
    double xh = lo_neg ? hi : lo;
    double x = 0.5 * (lo + hi);
    double dxold = fabs(hi - lo);
    double dx = dxold;
    double f = horner_t<N>(P, x);
    double df = horner_t<N - 1>(dP, x);
    for (int it = 0; it < 300; ++it) {
        if (f == 0.0) return x;
        if (f < 0.0) xl = x; else xh = x;
        const bool bisect = (((x - xh) * df - f) * ((x - xl) * df - f) > 0.0) ||
                            (fabs(2.0 * f) > fabs(dxold * df));
        dxold = dx;
        double xn;
        if (bisect) {
            dx = 0.5 * (xh - xl);
            xn = xl + dx;
            if (xn == xl || xn == xh) return x;
        } else {
            dx = f / df;
            xn = x - dx;
            if (xn == x) return x;
        }
        x = xn;
        f = horner_t<N>(P, x);
        df = horner_t<N - 1>(dP, x);
    }
    return x;
}


__device__ __forceinline__ double clamp_b(double v, double B)
{
    if (v < -B) v = -B;
    if (v > B) v = B;
    return v;
}

// One interval of one level of the root isolation: P has degree K, dP is its
// derivative; appends at most one root to cur[0..2].
template <int K>
__device__ __forceinline__ void root_interval(const double* P, const double* dP, double lo, double hi, double B,
                                              bool last, bool open_lo, bool open_hi, double* cur, int& mcur)
{
    lo = clamp_b(lo, B);
    hi = clamp_b(hi, B);
    if (!(hi > lo)) return;
    double flo = horner_t<K>(P, lo);
    double fhi = horner_t<K>(P, hi);
    tighten_open_interval([&](double x) { return horner_t<K>(P, x); }, open_lo, open_hi, lo, hi, flo, fhi);
    bool has = false;
    double r = 0.0;
    if (flo == 0.0) {
        has = true;
        r = lo;
    } else if (fhi != 0.0 && ((flo < 0.0) != (fhi < 0.0))) {
        has = true;
        r = solve_bracket_t<K>(P, dP, lo, hi, flo < 0.0);
    } else if (fhi == 0.0 && last) {
        has = true;
        r = hi;
    }
    if (has) {
        if (mcur == 0) cur[0] = r;
        else if (mcur == 1) cur[1] = r;
        else cur[2] = r;
        ++mcur;
    }
}

// Real roots, ascending, of q[0] + q[1] x + q[2] x^2, q[2] != 0: cancellation-free
// quadratic formula in unfused arithmetic with a correctly rounded square root;
// the double root of a zero discriminant is reported once.
__device__ __forceinline__ int quadratic_roots_t(const double* q, double* r)
{
    const double a = q[2], b = q[1], c = q[0];
    const double disc = b * b - 4.0 * a * c;
    if (!(disc >= 0.0)) return 0;
    if (disc == 0.0) {
        r[0] = -b / (2.0 * a);
        return 1;
    }
    const double s = sqrt(disc);
    const double t = -0.5 * (b + (b < 0.0 ? -s : s));
    const double r1 = t / a, r2 = c / t;
    r[0] = r1 < r2 ? r1 : r2;
    r[1] = r1 < r2 ? r2 : r1;
    return 2;
}

// Part k (0, 1, 2) of the real roots of cin[0..3] (any coefficient may vanish):
// a cubic takes its critical points from the closed form and part k is the
// interval between the (k-1)-th and the k-th of them, solved on its own; a
// quadratic hands out its two closed-form roots, a linear polynomial its one.
// The parts that report a root, taken in order of k, are the roots ascending.
__device__ __forceinline__ bool real_root3_part(const double* cin, int k, double& r)
{
    double cmax = 0.0;
#pragma unroll
    for (int i = 0; i <= 3; ++i) cmax = fmax(cmax, fabs(cin[i]));
    if (!(cmax > 0.0) || !(cmax <= DBL_MAX)) return false;
    double c[4];
#pragma unroll
    for (int i = 0; i <= 3; ++i) c[i] = cin[i] / cmax;
    if (c[3] != 0.0) {
        const double D2[3] = {1.0 * c[1], 2.0 * c[2], 3.0 * c[3]};
        double B = 0.0;
#pragma unroll
        for (int i = 0; i < 3; ++i) B = fmax(B, fabs(c[i] / c[3]));
        B = 1.0 + B;
        if (!(B < 1e25)) B = 1e25;
        double prev[2] = {0.0, 0.0};
        const int mprev = quadratic_roots_t(D2, prev);  // level 2: roots of the first derivative
        if (k > mprev) return false;
        // level 3: interval k of the polynomial itself
        const double lo = k == 0 ? -B : (k == 1 ? prev[0] : prev[1]);
        const double hi = k < mprev ? (k == 0 ? prev[0] : prev[1]) : B;
        double cur[3] = {0.0, 0.0, 0.0};
        int mcur = 0;
        root_interval<3>(c, D2, lo, hi, B, k == mprev, k == 0 && mprev >= 1, k == mprev && mprev >= 1, cur, mcur);
        r = cur[0];
        return mcur > 0;
    }
    if (c[2] != 0.0) {
        double q[2] = {0.0, 0.0};
        const int nq = quadratic_roots_t(c, q);
        r = k == 0 ? q[0] : q[1];
        return k < nq;
    }
    if (c[1] != 0.0) {
        r = -c[0] / c[1];
        return k == 0;
    }
    return false;
}

// ---- five-point solver, second half ---------------------------------------------
// Second half: real roots of the degree-10 polynomial and one model per root.
// FOUR hypotheses per wavefront, 16 lanes each: a level of the root isolation
// has at most 11 intervals, one per lane, so a wavefront given to a single
// hypothesis leaves 53 lanes idle through the longest part of the solver.
// Level k (roots of the (n-k)-th derivative) is spelled out per degree, with
// the coefficients of the level in registers.
struct Roots5Ws {
    double D[4][11][12];  // per hypothesis: D[k][i], coefficient i of the derivative of degree k
    double prev[4][12];
    double cur[4][12];
};

template <int K>
__device__ __forceinline__ void roots5_level(Roots5Ws& w, int g, int sl, bool live, int n, double B, int& mprev)
{
    bool has = false;
    double r = 0.0;
    const bool run = live && K <= n;  // a polynomial whose leading terms vanish stops at its own degree
    if (run && sl <= mprev) {
        double P[K + 1], dP[K];
#pragma unroll
        for (int i = 0; i <= K; ++i) P[i] = w.D[g][K][i];
#pragma unroll
        for (int i = 0; i < K; ++i) dP[i] = w.D[g][K - 1][i];
        const int t = sl;
        const double lo = (t == 0) ? -B : w.prev[g][t - 1];
        const double hi = (t < mprev) ? w.prev[g][t] : B;
        double cur[3] = {0.0, 0.0, 0.0};
        int mcur = 0;
        root_interval<K>(P, dP, lo, hi, B, t == mprev, t == 0 && mprev >= 1, t == mprev && mprev >= 1, cur, mcur);
        has = mcur > 0;
        r = cur[0];
    }
    const unsigned mine = (unsigned)(__builtin_amdgcn_ballot_w64(has) >> (16 * g)) & 0xFFFFu;
    __syncthreads();
    if (has) w.cur[g][__popc(mine & ((1u << sl) - 1u))] = r;
    __syncthreads();
    if (run) {
        mprev = __popc(mine);
        if (sl < mprev) w.prev[g][sl] = w.cur[g][sl];
    }
    __syncthreads();
}

__global__ void __launch_bounds__(64) ransac_roots5_kernel(const RansacProb* __restrict__ probs, int hyp0, int n_hyp)
{
    __shared__ Roots5Ws w;
    const RansacProb pb = probs[blockIdx.y];
    const int lane = threadIdx.x, g = lane >> 4, sl = lane & 15;
    const int t = blockIdx.x * 4 + g;
    const int hyp = hyp0 + t;
    const bool valid = t < n_hyp && pb.nmodels[hyp] == -1;
    double* slot = pb.models + (size_t)(valid ? hyp : hyp0) * 90;

    // scaled coefficients, degree after dropping vanishing leading terms, root bound
    double cmax = 0.0;
    double cn[11];
#pragma unroll
    for (int i = 0; i <= 10; ++i) {
        cn[i] = valid ? slot[FP_C + i] : 0.0;
        cmax = fmax(cmax, fabs(cn[i]));
    }
    bool live = valid && cmax > 0.0 && cmax <= DBL_MAX;
    int n = 10;
    double B = 0.0;
    if (live) {
#pragma unroll
        for (int i = 0; i <= 10; ++i) cn[i] = cn[i] / cmax;
        while (n > 0 && cn[n] == 0.0) --n;
        if (n == 0) live = false;
    }
    if (live) {
        double lead = cn[0];
#pragma unroll
        for (int i = 1; i <= 10; ++i) lead = (i == n) ? cn[i] : lead;
#pragma unroll
        for (int i = 0; i < 10; ++i)
            if (i < n) B = fmax(B, fabs(cn[i] / lead));
        B = 1.0 + B;
        if (!(B < 1e25)) B = 1e25;
#pragma unroll
        for (int i = 0; i <= 10; ++i)
            if (sl == i && i <= n) w.D[g][n][i] = cn[i];
    }
    __syncthreads();
    // derivative chain, one rounding per step as in the sequential statement
    for (int k = 10; k > 1; --k) {
        if (live && k <= n && sl < k) w.D[g][k - 1][sl] = (double)(sl + 1) * w.D[g][k][sl + 1];
        __syncthreads();
    }
    int mprev = 1;
    if (live && sl == 0) w.prev[g][0] = -w.D[g][1][0] / w.D[g][1][1];
    __syncthreads();
    roots5_level<2>(w, g, sl, live, n, B, mprev);
    roots5_level<3>(w, g, sl, live, n, B, mprev);
    roots5_level<4>(w, g, sl, live, n, B, mprev);
    roots5_level<5>(w, g, sl, live, n, B, mprev);
    roots5_level<6>(w, g, sl, live, n, B, mprev);
    roots5_level<7>(w, g, sl, live, n, B, mprev);
    roots5_level<8>(w, g, sl, live, n, B, mprev);
    roots5_level<9>(w, g, sl, live, n, B, mprev);
    roots5_level<10>(w, g, sl, live, n, B, mprev);
    const int nr = live ? mprev : 0;

    // one lane per root: (x, y) from the null vector of B(z), the model, its norm
    bool ok = false;
    double E[9];
    if (sl < nr) {
        const double z = w.prev[g][sl];
        double b[3][3];
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            double c0[4], c1[4], c2[5];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                c0[i] = slot[FP_B + 13 * r + i];
                c1[i] = slot[FP_B + 13 * r + 4 + i];
            }
#pragma unroll
            for (int i = 0; i < 5; ++i) c2[i] = slot[FP_B + 13 * r + 8 + i];
            b[r][0] = horner_t<3>(c0, z);
            b[r][1] = horner_t<3>(c1, z);
            b[r][2] = horner_t<4>(c2, z);
        }
        double bn0 = 0, bn1 = 0, bn2 = 0, bestz = -1.0;
        const int pr[3][2] = {{0, 1}, {0, 2}, {1, 2}};
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            const double* r0 = b[pr[q][0]];
            const double* r1 = b[pr[q][1]];
            const double nx = r0[1] * r1[2] - r0[2] * r1[1];
            const double ny = r0[2] * r1[0] - r0[0] * r1[2];
            const double nz = r0[0] * r1[1] - r0[1] * r1[0];
            const double az = fabs(nz);
            if (az > bestz) { bestz = az; bn0 = nx; bn1 = ny; bn2 = nz; }
        }
        const double nn = sqrt((bn0 * bn0 + bn1 * bn1) + bn2 * bn2);
        if (nn > 0.0 && (bestz / nn >= 1e-10)) {
            const double x = bn0 / bn2;
            const double y = bn1 / bn2;
            double s = 0.0;
#pragma unroll
            for (int i = 0; i < 9; ++i) {
                E[i] = ((x * slot[FP_N + i] + y * slot[FP_N + 9 + i]) + z * slot[FP_N + 18 + i]) + slot[FP_N + 27 + i];
                s = s + E[i] * E[i];
            }
            const double nrm = sqrt(s);
            if (nrm > 0.0 && nrm <= DBL_MAX) {
                ok = true;
#pragma unroll
                for (int i = 0; i < 9; ++i) E[i] = E[i] / nrm;
            }
        }
    }
    // every lane has read what it needs from the slot; the models overwrite it
    const unsigned mine = (unsigned)(__builtin_amdgcn_ballot_w64(ok) >> (16 * g)) & 0xFFFFu;
    __syncthreads();
    if (ok) {
        double* dst = slot + 9 * __popc(mine & ((1u << sl) - 1u));
#pragma unroll
        for (int i = 0; i < 9; ++i) dst[i] = E[i];
    }
    if (valid && sl == 0) pb.nmodels[hyp] = __popc(mine);
}

#define SFM_SA(i, j) sA[((i) * 9 + (j)) * 64 + lane]

// Seven-point fundamental matrix solver, one thread, first half.  m1/m2: the 7
// sampled float32 pixel points.  Leaves the pencil f1, f2 (the models are
// lambda f1 + f2 with det = 0); false when the system has rank < 7.
__device__ bool seven_point_pencil(double* sA, int lane, const float* m1, const float* m2, double* f1, double* f2)
{
#pragma unroll
    for (int i = 0; i < 7; ++i) {
        const double x0 = m1[2 * i], y0 = m1[2 * i + 1], x1 = m2[2 * i], y1 = m2[2 * i + 1];
        SFM_SA(i, 0) = x1 * x0; SFM_SA(i, 1) = x1 * y0; SFM_SA(i, 2) = x1;
        SFM_SA(i, 3) = y1 * x0; SFM_SA(i, 4) = y1 * y0; SFM_SA(i, 5) = y1;
        SFM_SA(i, 6) = x0;      SFM_SA(i, 7) = y0;      SFM_SA(i, 8) = 1.0;
    }
    // Gauss-Jordan with complete pivoting (first maximum in row-major scan);
    // perm holds the column permutation as 4-bit fields
    unsigned long long perm = 0x876543210ull;
    bool ok = true;
#pragma unroll
    for (int p = 0; p < 7; ++p) {
        int bi = p, bj = p;
        double best = -1.0;
#pragma unroll
        for (int i = p; i < 7; ++i)
#pragma unroll
            for (int j = p; j < 9; ++j) {
                const double v = fabs(SFM_SA(i, j));
                if (v > best) { best = v; bi = i; bj = j; }
            }
        if (!(best > 0.0)) ok = false;
        // swaps are carried out even when bi == p / bj == p (then they change nothing)
#pragma unroll
        for (int j = 0; j < 9; ++j) {
            const double a = SFM_SA(p, j), b = SFM_SA(bi, j);
            SFM_SA(bi, j) = a;
            SFM_SA(p, j) = b;
        }
#pragma unroll
        for (int i = 0; i < 7; ++i) {
            const double a = SFM_SA(i, p), b = SFM_SA(i, bj);
            SFM_SA(i, bj) = a;
            SFM_SA(i, p) = b;
        }
        {
            const unsigned long long np = (perm >> (4 * p)) & 15ull, nb = (perm >> (4 * bj)) & 15ull;
            perm &= ~((15ull << (4 * p)) | (15ull << (4 * bj)));
            perm |= (nb << (4 * p)) | (np << (4 * bj));
        }
        const double piv = SFM_SA(p, p);
        double rowp[9];
#pragma unroll
        for (int j = p + 1; j < 9; ++j) {
            rowp[j] = SFM_SA(p, j) / piv;
            SFM_SA(p, j) = rowp[j];
        }
        SFM_SA(p, p) = 1.0;
#pragma unroll
        for (int i = 0; i < 7; ++i) {
            if (i == p) continue;
            const double f = SFM_SA(i, p);
#pragma unroll
            for (int j = p + 1; j < 9; ++j) SFM_SA(i, j) = SFM_SA(i, j) - f * rowp[j];
            SFM_SA(i, p) = 0.0;
        }
    }
    if (!ok) return false;
    // basis vectors N0, N1 scattered through the permutation into the (now
    // unused) first seven columns: N[k][c] at element (k*9 + c) -> slot (t/7, t%7)
    double nv[2][7];
#pragma unroll
    for (int i = 0; i < 7; ++i) {
        nv[0][i] = -SFM_SA(i, 7);
        nv[1][i] = -SFM_SA(i, 8);
    }
#define SFM_SN(t) sA[((((t) / 7) * 9) + ((t) % 7)) * 64 + lane]
#pragma unroll
    for (int k = 0; k < 2; ++k) {
#pragma unroll
        for (int jj = 0; jj < 9; ++jj) {
            const int col = (int)((perm >> (4 * jj)) & 15ull);
            const double v = jj < 7 ? nv[k][jj < 7 ? jj : 0] : (jj == 7 + k ? 1.0 : 0.0);
            const int t = k * 9 + col;
            sA[(((t / 7) * 9) + (t % 7)) * 64 + lane] = v;
        }
    }
#pragma unroll
    for (int i = 0; i < 9; ++i) {
        f2[i] = SFM_SN(9 + i);
        f1[i] = SFM_SN(i) - f2[i];
    }
#undef SFM_SN
    return true;
}
#undef SFM_SA

// Second half, spread over three threads per hypothesis: the cubic
// det(lambda f1 + f2) into c[0..3] ...
__device__ __forceinline__ void seven_point_cubic(const double* f1, const double* f2, double* c)
{
    // entries are degree-1 polynomials p[i] = {f2[i], f1[i]}
    double pp[9][2];
#pragma unroll
    for (int i = 0; i < 9; ++i) { pp[i][0] = f2[i]; pp[i][1] = f1[i]; }
    double a[3], b[3], m0[3], mm1[3], mm2[3], t0[4], t1[4], t2[4];
    pmul_t<1, 1>(pp[4], pp[8], a); pmul_t<1, 1>(pp[5], pp[7], b);
#pragma unroll
    for (int i = 0; i < 3; ++i) m0[i] = a[i] - b[i];
    pmul_t<1, 1>(pp[3], pp[8], a); pmul_t<1, 1>(pp[5], pp[6], b);
#pragma unroll
    for (int i = 0; i < 3; ++i) mm1[i] = a[i] - b[i];
    pmul_t<1, 1>(pp[3], pp[7], a); pmul_t<1, 1>(pp[4], pp[6], b);
#pragma unroll
    for (int i = 0; i < 3; ++i) mm2[i] = a[i] - b[i];
    pmul_t<1, 2>(pp[0], m0, t0);
    pmul_t<1, 2>(pp[1], mm1, t1);
    pmul_t<1, 2>(pp[2], mm2, t2);
#pragma unroll
    for (int i = 0; i < 4; ++i) c[i] = (t0[i] - t1[i]) + t2[i];
}

// ... and the model of one root lambda (F33 = 1 when possible).
__device__ __forceinline__ void seven_point_model(const double* f1, const double* f2, double lambda, double* F)
{
    double mu = 1.0;
    const double s = f1[8] * lambda + f2[8];
    if (fabs(s) > DBL_EPSILON) {
        mu = 1.0 / s;
        lambda = lambda * mu;
        F[8] = 1.0;
    } else {
        F[8] = 0.0;
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) F[i] = f1[i] * lambda + f2[i] * mu;
}

// ---- error functions ---------------------------------------------------------

// Sampson error test: (float)(s^2 / denom) <= tf, with the division skipped
// when the outcome is certain (margins 2^-20 relative, far wider than any
// rounding involved); identical result to always dividing.
__device__ __forceinline__ int sampson_inlier(const double* E, double u1, double v1, double u2,
                                              double v2, float tf, double tlo, double thi)
{
    const double Ex1_0 = (E[0] * u1 + E[1] * v1) + E[2];
    const double Ex1_1 = (E[3] * u1 + E[4] * v1) + E[5];
    const double Ex1_2 = (E[6] * u1 + E[7] * v1) + E[8];
    const double Etx2_0 = (E[0] * u2 + E[3] * v2) + E[6];
    const double Etx2_1 = (E[1] * u2 + E[4] * v2) + E[7];
    const double s = (u2 * Ex1_0 + v2 * Ex1_1) + Ex1_2;
    const double a = Ex1_0 * Ex1_0;
    const double b = Ex1_1 * Ex1_1;
    const double c = Etx2_0 * Etx2_0;
    const double d = Etx2_1 * Etx2_1;
    const double den = ((a + b) + c) + d;
    const double s2 = s * s;
    if (den > 0.0 && den <= DBL_MAX) {
        if (s2 <= den * tlo) return 1;
        if (s2 >= den * thi) return 0;
    }
    return (float)(s2 / den) <= tf ? 1 : 0;
}

// Symmetric epipolar test: (float)max(d1^2/den1, d2^2/den2) <= tf, evaluated as
// OpenCV does (reciprocal, then two multiplies) only when the outcome is not
// already certain from den*t comparisons with 2^-20 relative margins.
__device__ __forceinline__ int symepi_inlier_d(const double* F, double x1, double y1, double x2, double y2,
                                               float tf, double tlo, double thi)
{
    double a = (F[0] * x1 + F[1] * y1) + F[2];
    double b = (F[3] * x1 + F[4] * y1) + F[5];
    double c = (F[6] * x1 + F[7] * y1) + F[8];
    const double den2 = a * a + b * b;
    const double d2 = (x2 * a + y2 * b) + c;
    a = (F[0] * x2 + F[3] * y2) + F[6];
    b = (F[1] * x2 + F[4] * y2) + F[7];
    c = (F[2] * x2 + F[5] * y2) + F[8];
    const double den1 = a * a + b * b;
    const double d1 = (x1 * a + y1 * b) + c;
    const double n1 = d1 * d1, n2 = d2 * d2;
    if (den1 > 0.0 && den1 <= DBL_MAX && den2 > 0.0 && den2 <= DBL_MAX) {
        if (n1 >= den1 * thi || n2 >= den2 * thi) return 0;
        if (n1 <= den1 * tlo && n2 <= den2 * tlo) return 1;
    }
    const double s2 = 1. / den2;
    const double s1 = 1. / den1;
    return (float)fmax(d1 * d1 * s1, d2 * d2 * s2) <= tf ? 1 : 0;
}
__device__ __forceinline__ int symepi_inlier(const double* F, float fx1, float fy1, float fx2, float fy2,
                                             float tf, double tlo, double thi)
{
    return symepi_inlier_d(F, (double)fx1, (double)fy1, (double)fx2, (double)fy2, tf, tlo, thi);
}

// ---- kernels -------------------------------------------------------------------

// Prepares the points of each problem from its pixel coordinates
// (ransac_prep_point) and clears the per-problem result.
__global__ void __launch_bounds__(256) ransac_prep_kernel(const RansacProb* __restrict__ probs)
{
    const RansacProb pb = probs[blockIdx.y];
    const int n = *pb.n_ptr;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i == 0) {
        ransac_reset_result(pb);
        pb.result[R_N] = n;
    }
    if (i >= n) return;
    ransac_prep_point(pb, i, pb.px1[2 * i], pb.px1[2 * i + 1], pb.px2[2 * i], pb.px2[2 * i + 1]);
}

struct CvRng {
    unsigned long long state;
    __device__ unsigned next()
    {
        state = (unsigned long long)(unsigned)state * 4164903690U + (unsigned)(state >> 32);
        return (unsigned)state;
    }
    __device__ int uniform(int a, int b) { return a == b ? a : (int)(next() % (unsigned)(b - a) + a); }
};

// OpenCV's subset check for the fundamental matrix: is the last point of the
// subset (nearly) collinear with any two earlier ones.  p holds the subset's
// points (x,y interleaved).
__device__ bool have_collinear(const float* p, int count)
{
    const int i = count - 1;
    for (int j = 0; j < i; ++j) {
        const double dx1 = p[2 * j] - p[2 * i];
        const double dy1 = p[2 * j + 1] - p[2 * i + 1];
        for (int k = 0; k < j; ++k) {
            const double dx2 = p[2 * k] - p[2 * i];
            const double dy2 = p[2 * k + 1] - p[2 * i + 1];
            if (fabs(dx2 * dy1 - dy2 * dx1) <=
                FLT_EPSILON * (fabs(dx1) + fabs(dy1) + fabs(dx2) + fabs(dy2)))
                return true;
        }
    }
    return false;
}

__device__ bool subset_collinear(const RansacProb& pb, const int* idx, int m)
{
    float a[14], b[14];
    for (int k = 0; k < m; ++k) {
        const float2 u = ((const float2*)pb.f1)[idx[k]];
        const float2 v = ((const float2*)pb.f2)[idx[k]];
        a[2 * k] = u.x; a[2 * k + 1] = u.y;
        b[2 * k] = v.x; b[2 * k + 1] = v.y;
    }
    return have_collinear(a, m) || have_collinear(b, m);
}

// The same check for a seven-point subset with every gather issued up front.
__device__ bool subset_collinear7(const RansacProb& pb, const int* idx)
{
    float a[14], b[14];
#pragma unroll
    for (int k = 0; k < 7; ++k) {
        const float2 u = ((const float2*)pb.f1)[idx[k]];
        const float2 v = ((const float2*)pb.f2)[idx[k]];
        a[2 * k] = u.x; a[2 * k + 1] = u.y;
        b[2 * k] = v.x; b[2 * k + 1] = v.y;
    }
    return have_collinear(a, 7) || have_collinear(b, 7);
}

// Fixed-count mode: rows are independent draws of a counter-based generator
// (splitmix64 finaliser over row / subset attempt / slot / slot attempt), one
// thread per row.  No OpenCV counterpart; the adaptive mode below keeps
// OpenCV's sequential stream.
__device__ __forceinline__ unsigned long long mix64(unsigned long long z)
{
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}

__device__ __forceinline__ int fixed_draw(int count, unsigned row, unsigned u, unsigned i, unsigned t)
{
    const unsigned long long ctr = ((unsigned long long)row << 32) | ((unsigned long long)(u & 0xFF) << 24) |
                                   ((unsigned long long)(i & 0xFF) << 16) | (unsigned long long)(t & 0xFFFF);
    const unsigned long long h = mix64(0x53464D5F5441424CULL + ctr * 0x9E3779B97F4A7C15ULL);
    return (int)(((h >> 32) * (unsigned long long)(unsigned)count) >> 32);
}

__global__ void __launch_bounds__(256) ransac_table_fixed_kernel(const RansacProb* __restrict__ probs, int nrows)
{
    const RansacProb pb = probs[blockIdx.y];
    const int n = *pb.n_ptr;
    const int m = pb.m;
    const int r = blockIdx.x * 256 + threadIdx.x;
    if (n < m) return;
    if (n == m) {
        if (r == 0) {
            for (int i = 0; i < m; ++i) pb.table[i] = i;
            pb.result[R_VALID] = 1;
            pb.result[R_DIRECT] = 1;
        }
        return;
    }
    if (r == 0) pb.result[R_VALID] = nrows;
    if (r >= nrows) return;
    const bool check = pb.model == SFM_GPU_MODEL_FUNDAMENTAL;
    int idx[7];
    bool ok = false;
    for (unsigned u = 0; u < 64 && !ok; ++u) {
        for (int i = 0; i < m; ++i) {
            for (unsigned t = 0;; ++t) {
                const int v = fixed_draw(n, (unsigned)r, u, (unsigned)i, t);
                int j = 0;
                for (; j < i; ++j)
                    if (idx[j] == v) break;
                if (j == i) { idx[i] = v; break; }
            }
        }
        ok = !(check && subset_collinear7(pb, idx));  // check implies m == 7
    }
    if (!ok) idx[0] = -1;
    for (int k = 0; k < m; ++k) pb.table[(size_t)r * m + k] = idx[k];
}

// Sample table rows [row0, row0+nrows): m distinct uniform draws per row from
// the cv::RNG stream seeded with 2^64-1, whole-subset redraw when the
// fundamental-matrix collinearity check rejects it (a row fails after 10000
// rejections in a row).  The stream of candidate subsets does not depend on
// which of them are accepted, so lane 0 draws a run of candidates (pure
// arithmetic), all lanes check one candidate each, and the accepted ones are
// taken in stream order.  The generator state after the last candidate used is
// parked in the result block between calls.
__global__ void __launch_bounds__(64) ransac_table_kernel(const RansacProb* __restrict__ probs, int row0, int nrows)
{
    __shared__ int cand[64][8];
    __shared__ unsigned long long st_after[64];
    const RansacProb pb = probs[blockIdx.x];
    const int lane = threadIdx.x;
    const int n = *pb.n_ptr;
    const int m = pb.m;
    if (n < m) return;
    if (row0 > 0 && pb.result[R_VALID] < row0) return;  // stream ended earlier
    if (n == m) {
        if (row0 == 0 && lane == 0) {
            for (int i = 0; i < m; ++i) pb.table[i] = i;
            pb.result[R_VALID] = 1;
            pb.result[R_DIRECT] = 1;
        }
        return;
    }
    unsigned long long state;
    if (row0 == 0) state = ~0ull;
    else state = ((unsigned long long)(unsigned)pb.result[R_RNG_HI] << 32) | (unsigned)pb.result[R_RNG_LO];
    const bool check = pb.model == SFM_GPU_MODEL_FUNDAMENTAL;
    // x % n for 32-bit x without a division: r = hi64((M * x) * n), M = floor((2^64-1)/n) + 1
    const unsigned long long magic = 0xFFFFFFFFFFFFFFFFull / (unsigned)n + 1ull;
    const int end = row0 + nrows;
    int valid = row0;
    int rejects = 0;  // consecutive rejected candidates for the row being filled
    bool failed = false;
    while (valid < end && !failed) {
        const int chunk = min(64, end - valid);
        if (lane == 0) {
            CvRng rng;
            rng.state = state;
            for (int c = 0; c < chunk; ++c) {
                int idx[7];
#pragma unroll
                for (int i = 0; i < 7; ++i) {
                    if (i < m) {
                        int v;
                        for (;;) {
                            const unsigned x = rng.next();
                            v = (int)__umul64hi(magic * x, (unsigned long long)(unsigned)n);
                            bool dup = false;
#pragma unroll
                            for (int j = 0; j < i; ++j) dup |= (idx[j] == v);
                            if (!dup) break;
                        }
                        idx[i] = v;
                        cand[c][i] = v;
                    }
                }
                st_after[c] = rng.state;
            }
        }
        __syncthreads();
        bool rej = false;
        if (check && lane < chunk) {
            int idx[7];
#pragma unroll
            for (int i = 0; i < 7; ++i) idx[i] = cand[lane][i];
            rej = subset_collinear7(pb, idx);
        }
        const unsigned long long rejmask = __builtin_amdgcn_ballot_w64(rej);
        int last = -1;
        for (int c = 0; c < chunk && valid < end; ++c) {
            last = c;
            if ((rejmask >> c) & 1ull) {
                if (++rejects >= 10000) {
                    failed = true;
                    break;
                }
            } else {
                if (lane < m) pb.table[(size_t)valid * m + lane] = cand[c][lane];
                ++valid;
                rejects = 0;
            }
        }
        state = st_after[last];
        __syncthreads();
    }
    if (lane == 0) {
        pb.result[R_VALID] = valid;
        pb.result[R_RNG_HI] = (int)(unsigned)(state >> 32);
        pb.result[R_RNG_LO] = (int)(unsigned)state;
    }
}

// Five-point hypotheses, one wavefront each.
__global__ void __launch_bounds__(64) ransac_solve_kernel(const RansacProb* __restrict__ probs, int hyp0)
{
    __shared__ SolveWs ws;
    __shared__ double sx1[14], sx2[14];
    const RansacProb pb = probs[blockIdx.y];
    const int hyp = hyp0 + blockIdx.x;
    const int lane = threadIdx.x;
    const int n = *pb.n_ptr;
    const int m = pb.m;
    if (n < m || hyp >= pb.result[R_VALID]) {
        if (lane == 0) pb.nmodels[hyp] = 0;
        return;
    }
    const int* row = pb.table + (size_t)hyp * m;
    if (row[0] < 0) {  // the row could not be drawn
        if (lane == 0) pb.nmodels[hyp] = 0;
        return;
    }
    if (lane < 2 * m) {
        const int pt = row[lane >> 1], comp = lane & 1;
        sx1[lane] = pb.x1[2 * pt + comp];
        sx2[lane] = pb.x2[2 * pt + comp];
    }
    __syncthreads();
    double* slot = pb.models + (size_t)hyp * 90;
    const bool ok = five_point_wave(ws, sx1, sx2, slot, lane);
    if (lane == 0) pb.nmodels[hyp] = ok ? -1 : 0;  // -1: waiting for ransac_roots5_kernel
}

// Seven-point hypotheses, one thread each, in two kernels: the elimination
// needs 32 KB of LDS per wavefront (five wavefronts per CU), the root finding
// that follows needs none and runs at full occupancy.  The pencil travels
// through the hypothesis' own slot of the models buffer; nmodels = -1 marks a
// hypothesis whose pencil is waiting there.
// gen_rows > 0 (fixed-count mode, hyp0 = 0): the kernel draws the sample rows
// itself, exactly as ransac_table_fixed_kernel would (the points a row's
// collinearity check gathers are the ones the solver needs next), stores them
// in the table for the test hooks and sets the row count; gen_rows = 0: the
// rows are read from the table.
__global__ void __launch_bounds__(64) ransac_solve7_kernel(const RansacProb* __restrict__ probs, int hyp0, int n_hyp,
                                                           int gen_rows)
{
    __shared__ double sA[63 * 64];
    const RansacProb pb = probs[blockIdx.y];
    const int lane = threadIdx.x;
    const int hyp = hyp0 + blockIdx.x * 64 + lane;
    const bool in_range = hyp < hyp0 + n_hyp;
    const int n = *pb.n_ptr;
    const int valid = gen_rows > 0 ? (n == 7 ? 1 : gen_rows) : pb.result[R_VALID];
    bool active = in_range && n >= 7 && hyp < valid;
    int idx[7];
#pragma unroll
    for (int k = 0; k < 7; ++k) idx[k] = 0;
    float m1[14], m2[14];
    auto gather = [&]() {
#pragma unroll
        for (int k = 0; k < 7; ++k) {
            const float2 u = ((const float2*)pb.f1)[idx[k]];
            const float2 v = ((const float2*)pb.f2)[idx[k]];
            m1[2 * k] = u.x; m1[2 * k + 1] = u.y;
            m2[2 * k] = v.x; m2[2 * k + 1] = v.y;
        }
    };
    if (gen_rows > 0) {
        if (n >= 7 && blockIdx.x == 0 && lane == 0) {
            pb.result[R_VALID] = valid;
            if (n == 7) pb.result[R_DIRECT] = 1;
        }
        if (active) {
            if (n == 7) {
#pragma unroll
                for (int k = 0; k < 7; ++k) idx[k] = k;
                gather();
            } else {
                bool ok = false;
                for (unsigned u = 0; u < 64 && !ok; ++u) {
                    for (int i = 0; i < 7; ++i) {
                        for (unsigned t = 0;; ++t) {
                            const int v = fixed_draw(n, (unsigned)hyp, u, (unsigned)i, t);
                            int j = 0;
                            for (; j < i; ++j)
                                if (idx[j] == v) break;
                            if (j == i) { idx[i] = v; break; }
                        }
                    }
                    gather();
                    ok = !(have_collinear(m1, 7) || have_collinear(m2, 7));
                }
                if (!ok) idx[0] = -1;
            }
            int* row = pb.table + (size_t)hyp * 7;
#pragma unroll
            for (int k = 0; k < 7; ++k) row[k] = idx[k];
        }
    } else if (active) {
        const int* row = pb.table + (size_t)hyp * 7;
#pragma unroll
        for (int k = 0; k < 7; ++k) idx[k] = row[k];
    }
    if (active && idx[0] < 0) active = false;  // the row could not be drawn
    // inactive lanes run on seven copies of point 0 (rejected at the first
    // zero pivot) and drop the result
    if (!active) {
#pragma unroll
        for (int k = 0; k < 7; ++k) idx[k] = 0;
    }
    if (n > 0) {
        if (!(gen_rows > 0 && active)) gather();
    } else {
#pragma unroll
        for (int k = 0; k < 14; ++k) m1[k] = m2[k] = 0.f;
    }
    double f1[9], f2[9];
    const bool ok = seven_point_pencil(sA, lane, m1, m2, f1, f2);
    if (!in_range) return;
    if (!active || !ok) {
        pb.nmodels[hyp] = 0;
        return;
    }
    pb.nmodels[hyp] = -1;
    double* dst = pb.models + (size_t)hyp * 90;
#pragma unroll
    for (int i = 0; i < 9; ++i) {
        dst[i] = f1[i];
        dst[9 + i] = f2[i];
    }
}

// The same elimination with LPH lanes per hypothesis (64 / LPH hypotheses per
// wavefront, 63 * 8 * 64 / LPH bytes of LDS): the lanes of a hypothesis share its
// matrix in LDS and split every loop of a step among them by index modulo LPH --
// rows for the pivot search, the column swap and the elimination, columns for the
// row swap and the division of the pivot row -- so that all lanes run one
// instruction stream.  Every element goes through the operations of
// seven_point_pencil in the same order; only who carries them out differs.  A
// wavefront's LDS accesses are served in program order, so what one lane wrote is
// there for its partner's next read; the fences keep the compiler from moving
// accesses across those hand-overs.
template <int LPH>
__device__ __forceinline__ bool seven_point_pencil_split(double* sA, int h, int r, const float* m1, const float* m2,
                                                         double* dst, bool writes)
{
    constexpr int HPW = 64 / LPH;
#define SFM_SB2(i, j) sA[((i) * 9 + (j)) * HPW + h]
    auto handover = [] { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); };
#pragma unroll
    for (int k = 0; k < (7 + LPH - 1) / LPH; ++k) {
        const int i = LPH * k + r;
        if (i < 7) {
            // (the lane's rows: points selected without indexing the arrays by a lane-dependent value)
            float x0f = 0.f, y0f = 0.f, x1f = 0.f, y1f = 0.f;
#pragma unroll
            for (int q = 0; q < LPH; ++q)
                if (LPH * k + q < 7 && r == q) {
                    x0f = m1[2 * (LPH * k + q)];
                    y0f = m1[2 * (LPH * k + q) + 1];
                    x1f = m2[2 * (LPH * k + q)];
                    y1f = m2[2 * (LPH * k + q) + 1];
                }
            const double x0 = x0f, y0 = y0f, x1 = x1f, y1 = y1f;
            SFM_SB2(i, 0) = x1 * x0; SFM_SB2(i, 1) = x1 * y0; SFM_SB2(i, 2) = x1;
            SFM_SB2(i, 3) = y1 * x0; SFM_SB2(i, 4) = y1 * y0; SFM_SB2(i, 5) = y1;
            SFM_SB2(i, 6) = x0;      SFM_SB2(i, 7) = y0;      SFM_SB2(i, 8) = 1.0;
        }
    }
    handover();
    unsigned long long perm = 0x876543210ull;
    bool ok = true;
#pragma unroll
    for (int p = 0; p < 7; ++p) {
        // first maximum in row-major scan order: every lane over its rows, then across the lanes
        int bi = p, bj = p;
        double best = -1.0;
#pragma unroll
        for (int k = p / LPH; k < (7 + LPH - 1) / LPH; ++k) {
            const int i = LPH * k + r;
            const bool row_ok = i >= p && i < 7;
#pragma unroll
            for (int j = p; j < 9; ++j) {
                const double v = fabs(SFM_SB2(row_ok ? i : p, j));
                if (row_ok && v > best) { best = v; bi = i; bj = j; }
            }
        }
#pragma unroll
        for (int off = 1; off < LPH; off <<= 1) {
            const double ob = __shfl_xor(best, off);
            const int oi = __shfl_xor(bi, off), oj = __shfl_xor(bj, off);
            // (equal values in different rows: the earlier row comes first in the scan; a lane
            // that found nothing still holds (-1, p, p), like its partner then)
            if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; bj = oj; }
        }
        if (!(best > 0.0)) ok = false;
        // row swap p <-> bi, columns dealt out among the lanes
#pragma unroll
        for (int k = 0; k < (9 + LPH - 1) / LPH; ++k) {
            const int j = LPH * k + r;
            if (j < 9) {
                const double a = SFM_SB2(p, j), b = SFM_SB2(bi, j);
                SFM_SB2(bi, j) = a;
                SFM_SB2(p, j) = b;
            }
        }
        handover();
        // column swap p <-> bj, rows dealt out
#pragma unroll
        for (int k = 0; k < (7 + LPH - 1) / LPH; ++k) {
            const int i = LPH * k + r;
            if (i < 7) {
                const double a = SFM_SB2(i, p), b = SFM_SB2(i, bj);
                SFM_SB2(i, bj) = a;
                SFM_SB2(i, p) = b;
            }
        }
        {
            const unsigned long long np = (perm >> (4 * p)) & 15ull, nb = (perm >> (4 * bj)) & 15ull;
            perm &= ~((15ull << (4 * p)) | (15ull << (4 * bj)));
            perm |= (nb << (4 * p)) | (np << (4 * bj));
        }
        handover();
        // the pivot row divided by the pivot, columns dealt out
        const double piv = SFM_SB2(p, p);
        handover();  // (everybody has the pivot before it is overwritten)
#pragma unroll
        for (int k = 0; k < (8 - p + LPH - 1) / LPH; ++k) {
            const int j = p + 1 + LPH * k + r;
            if (j < 9) SFM_SB2(p, j) = SFM_SB2(p, j) / piv;
        }
        if (r == 0) SFM_SB2(p, p) = 1.0;
        handover();
        double rowp[9];
#pragma unroll
        for (int j = p + 1; j < 9; ++j) rowp[j] = SFM_SB2(p, j);
#pragma unroll
        for (int k = 0; k < (7 + LPH - 1) / LPH; ++k) {
            const int i = LPH * k + r;
            if (i < 7 && i != p) {
                const double f = SFM_SB2(i, p);
#pragma unroll
                for (int j = p + 1; j < 9; ++j) SFM_SB2(i, j) = SFM_SB2(i, j) - f * rowp[j];
                SFM_SB2(i, p) = 0.0;
            }
        }
        handover();
    }
    if (!ok) return false;
    // The two null-space vectors through the permutation: entry `col` of vector k is
    // -A[jj][7 + k] for jj < 7 and 1 / 0 for the two free columns, where perm[jj] = col.
    // Lane r delivers the pencil's entries i = r, r + LPH, ...: f2[i] = N1[i], f1[i] = N0[i] - N1[i].
#pragma unroll
    for (int k = 0; k < (9 + LPH - 1) / LPH; ++k) {
        const int i = LPH * k + r;
        if (i < 9) {
            double n0 = 0.0, n1 = 0.0;
#pragma unroll
            for (int jj = 0; jj < 9; ++jj) {
                const int col = (int)((perm >> (4 * jj)) & 15ull);
                if (col == i) {
                    n0 = jj < 7 ? -SFM_SB2(jj < 7 ? jj : 0, 7) : (jj == 7 ? 1.0 : 0.0);
                    n1 = jj < 7 ? -SFM_SB2(jj < 7 ? jj : 0, 8) : (jj == 8 ? 1.0 : 0.0);
                }
            }
            if (writes) {
                dst[i] = n0 - n1;
                dst[9 + i] = n1;
            }
        }
    }
#undef SFM_SB2
    return true;
}

template <int LPH>
__global__ void __launch_bounds__(64) ransac_solve7_split_kernel(const RansacProb* __restrict__ probs, int hyp0,
                                                                 int n_hyp, int gen_rows)
{
    constexpr int HPW = 64 / LPH;
    __shared__ double sA[63 * HPW];
    const RansacProb pb = probs[blockIdx.y];
    const int lane = threadIdx.x;
    const int h = lane / LPH, r = lane % LPH;
    const int hyp = hyp0 + blockIdx.x * HPW + h;
    const bool in_range = hyp < hyp0 + n_hyp;
    const int n = *pb.n_ptr;
    const int valid = gen_rows > 0 ? (n == 7 ? 1 : gen_rows) : pb.result[R_VALID];
    bool active = in_range && n >= 7 && hyp < valid;
    int idx[7];
#pragma unroll
    for (int k = 0; k < 7; ++k) idx[k] = 0;
    float m1[14], m2[14];
    auto gather = [&]() {
#pragma unroll
        for (int k = 0; k < 7; ++k) {
            const float2 u = ((const float2*)pb.f1)[idx[k]];
            const float2 v = ((const float2*)pb.f2)[idx[k]];
            m1[2 * k] = u.x; m1[2 * k + 1] = u.y;
            m2[2 * k] = v.x; m2[2 * k + 1] = v.y;
        }
    };
    // (the lanes of a hypothesis draw the same rows; each checks one of the two images for collinear points)
    if (gen_rows > 0) {
        if (n >= 7 && blockIdx.x == 0 && lane == 0) {
            pb.result[R_VALID] = valid;
            if (n == 7) pb.result[R_DIRECT] = 1;
        }
        if (active) {
            if (n == 7) {
#pragma unroll
                for (int k = 0; k < 7; ++k) idx[k] = k;
                gather();
            } else {
                bool ok = false;
                for (unsigned u = 0; u < 64 && !ok; ++u) {
                    for (int i = 0; i < 7; ++i) {
                        for (unsigned t = 0;; ++t) {
                            const int v = fixed_draw(n, (unsigned)hyp, u, (unsigned)i, t);
                            int j = 0;
                            for (; j < i; ++j)
                                if (idx[j] == v) break;
                            if (j == i) { idx[i] = v; break; }
                        }
                    }
                    gather();
                    float mm[14];
#pragma unroll
                    for (int k = 0; k < 14; ++k) mm[k] = (r & 1) ? m2[k] : m1[k];
                    bool bad = have_collinear(mm, 7);
                    bad = bad || __shfl_xor((int)bad, 1) != 0;
                    ok = !bad;
                }
                if (!ok) idx[0] = -1;
            }
            if (r == 0) {
                int* row = pb.table + (size_t)hyp * 7;
#pragma unroll
                for (int k = 0; k < 7; ++k) row[k] = idx[k];
            }
        }
    } else if (active) {
        const int* row = pb.table + (size_t)hyp * 7;
#pragma unroll
        for (int k = 0; k < 7; ++k) idx[k] = row[k];
    }
    if (active && idx[0] < 0) active = false;  // the row could not be drawn
    if (!active) {
#pragma unroll
        for (int k = 0; k < 7; ++k) idx[k] = 0;
    }
    if (n > 0) {
        if (!(gen_rows > 0 && active)) gather();
    } else {
#pragma unroll
        for (int k = 0; k < 14; ++k) m1[k] = m2[k] = 0.f;
    }
    double* dst = pb.models + (size_t)(in_range ? hyp : hyp0) * 90;
    const bool ok = seven_point_pencil_split<LPH>(sA, h, r, m1, m2, dst, in_range && active);
    if (!in_range || r != 0) return;
    pb.nmodels[hyp] = (active && ok) ? -1 : 0;
}

// Second half of the seven-point solver: the cubic det(lambda f1 + f2) and one model
// per real root.  Three lanes per hypothesis, one interval of the cubic each (the
// Newton loops of the three intervals run side by side and three times as many
// wavefronts hide their dependent double-precision chains), arranged as one
// wavefront per PART: a workgroup of three wavefronts takes 64 hypotheses,
// wavefront k their interval k.  The outer intervals (bracketed by doubling steps
// from their finite end, then Newton) take more iterations than the middle one,
// and a wavefront runs as long as its slowest lane: with the three parts of a
// hypothesis side by side in one wavefront (21 hypotheses per wavefront) every
// wavefront waited for its outer intervals (0.820 against 0.813 ms per bench step).
__global__ void __launch_bounds__(192) ransac_roots7_parts_kernel(const RansacProb* __restrict__ probs, int hyp0,
                                                                  int n_hyp)
{
    __shared__ unsigned char s_has[3][64];
    const RansacProb pb = probs[blockIdx.y];
    const int lane = threadIdx.x & 63, k = threadIdx.x >> 6;
    const int t = blockIdx.x * 64 + lane;
    const int hyp = hyp0 + t;
    // every wavefront of the workgroup reads the flag and the pencil before any of them writes
    const bool valid = t < n_hyp && pb.nmodels[hyp] == -1;
    double* dst = pb.models + (size_t)(valid ? hyp : hyp0) * 90;
    double f1[9], f2[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) {
        f1[i] = valid ? dst[i] : 0.0;
        f2[i] = valid ? dst[9 + i] : 0.0;
    }
    double c[4], r = 0.0;
    seven_point_cubic(f1, f2, c);
    const bool has = valid && real_root3_part(c, k, r);
    s_has[k][lane] = has ? 1 : 0;
    __syncthreads();
    const int h0 = s_has[0][lane], h1 = s_has[1][lane], h2 = s_has[2][lane];
    if (has) {
        double F[9];
        seven_point_model(f1, f2, r, F);
        const int pos = (k > 0 ? h0 : 0) + (k > 1 ? h1 : 0);
#pragma unroll
        for (int i = 0; i < 9; ++i) dst[9 * pos + i] = F[i];
    }
    if (valid && k == 0) pb.nmodels[hyp] = h0 + h1 + h2;
}

// Staged scoring: points every listed model is scored on before the stages are sized.
constexpr int SCORE_OPENING = 256;
// The rest of the first stage runs as this many equal pieces per problem (measured 2 / 4 / pieces of 256
// with idle workgroups beyond a problem's stage: 259 / 230 / 238 us at 25 % outliers, 82 / 77 / 93 us at 7 %).
constexpr int SCORE_SPLIT = 4;

// Lists the candidate models of hypotheses [hyp0, hyp0+n_hyp) densely
// (entry = hyp*10 + q, scan order) and initialises their counts: 0 for an
// existing model, -1 for none.  One workgroup per problem.
__global__ void __launch_bounds__(1024) ransac_compact_kernel(const RansacProb* __restrict__ probs, int hyp0,
                                                              int n_hyp, int maxm)
{
    __shared__ int wsum[16];
    __shared__ int s_running;
    const RansacProb pb = probs[blockIdx.x];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int n = *pb.n_ptr;
    const int valid = pb.result[R_VALID];
    if (tid == 0) s_running = 0;
    // largest |coordinate| over the problem's points, as float bits (non-negative
    // floats order like their bit patterns); the filter's error bounds use it
    {
        int mx = 0;
        // one 64-byte block (four points, four planes) per thread and turn, four turns' loads in flight
        const float4* blocks = (const float4*)pb.soa;
        const int nblk = (n + 3) / 4;
        for (int b0 = tid; b0 < nblk; b0 += 4 * 1024) {
            float4 v[4][4];
#pragma unroll
            for (int k = 0; k < 4; ++k)
#pragma unroll
                for (int c = 0; c < 4; ++c)
                    v[k][c] = b0 + 1024 * k < nblk ? blocks[(size_t)(b0 + 1024 * k) * 4 + c] : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int valid_pts = n - 4 * (b0 + 1024 * k);  // points of the block that exist (the rest is stale)
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    if (valid_pts > 0) mx = max(mx, __float_as_int(fabsf(v[k][c].x)));
                    if (valid_pts > 1) mx = max(mx, __float_as_int(fabsf(v[k][c].y)));
                    if (valid_pts > 2) mx = max(mx, __float_as_int(fabsf(v[k][c].z)));
                    if (valid_pts > 3) mx = max(mx, __float_as_int(fabsf(v[k][c].w)));
                }
            }
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) mx = max(mx, __shfl_xor(mx, off));
        if (lane == 0) wsum[wv] = mx;
        __syncthreads();
        if (tid == 0) {
            for (int w = 1; w < 16; ++w) mx = max(mx, wsum[w]);
            pb.result[R_MAXABS] = mx;
        }
    }
    __syncthreads();
    for (int h0 = 0; h0 < n_hyp; h0 += 1024) {
        const int hyp = hyp0 + h0 + tid;
        const bool in_range = h0 + tid < n_hyp;
        int nm = 0;
        if (in_range && n >= pb.m && hyp < valid) nm = min(pb.nmodels[hyp], maxm);
        int incl = nm;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const int o = __shfl_up(incl, off);
            if (lane >= off) incl += o;
        }
        if (lane == 63) wsum[wv] = incl;
        __syncthreads();
        int base = s_running;
        for (int w = 0; w < wv; ++w) base += wsum[w];
        const int pos = base + incl - nm;
        if (in_range) {
            for (int q = 0; q < MAX_MODELS; ++q) pb.counts[(size_t)hyp * MAX_MODELS + q] = q < nm ? 0 : -1;
            for (int q = 0; q < nm; ++q) pb.list[pos + q] = hyp * MAX_MODELS + q;
        }
        __syncthreads();
        if (tid == 1023) s_running = base + incl;
        __syncthreads();
    }
    if (tid == 0) {
        pb.result[R_NLIST] = s_running;
        pb.result[R_BOUND] = 0;
        pb.result[R_S0] = SCORE_OPENING;
    }
}

// den > 0 && den <= DBL_MAX in one instruction: positive normal or denormal
__device__ __forceinline__ bool pos_finite(double v) { return __builtin_amdgcn_class(v, 0x180); }

// The two tests again for the one-thread-per-model kernel: same arithmetic and
// same outcome as sampson_inlier / symepi_inlier_d, arranged so that the common
// case (outcome certain without dividing) has no branch.
__device__ __forceinline__ int sampson_inlier_u(const double* E, double u1, double v1, double u2, double v2,
                                                float tf, double tlo, double thi)
{
    const double Ex1_0 = (E[0] * u1 + E[1] * v1) + E[2];
    const double Ex1_1 = (E[3] * u1 + E[4] * v1) + E[5];
    const double Ex1_2 = (E[6] * u1 + E[7] * v1) + E[8];
    const double Etx2_0 = (E[0] * u2 + E[3] * v2) + E[6];
    const double Etx2_1 = (E[1] * u2 + E[4] * v2) + E[7];
    const double s = (u2 * Ex1_0 + v2 * Ex1_1) + Ex1_2;
    const double a = Ex1_0 * Ex1_0;
    const double b = Ex1_1 * Ex1_1;
    const double c = Etx2_0 * Etx2_0;
    const double d = Etx2_1 * Etx2_1;
    const double den = ((a + b) + c) + d;
    const double s2 = s * s;
    const bool in = s2 <= den * tlo;
    const bool out = s2 >= den * thi;
    int r = (in && !out) ? 1 : 0;
    if (!(pos_finite(den) && (in || out))) r = (float)(s2 / den) <= tf ? 1 : 0;
    return r;
}

__device__ __forceinline__ int symepi_inlier_u(const double* F, double x1, double y1, double x2, double y2,
                                               float tf, double tlo, double thi)
{
    double a = (F[0] * x1 + F[1] * y1) + F[2];
    double b = (F[3] * x1 + F[4] * y1) + F[5];
    double c = (F[6] * x1 + F[7] * y1) + F[8];
    const double den2 = a * a + b * b;
    const double d2 = (x2 * a + y2 * b) + c;
    a = (F[0] * x2 + F[3] * y2) + F[6];
    b = (F[1] * x2 + F[4] * y2) + F[7];
    c = (F[2] * x2 + F[5] * y2) + F[8];
    const double den1 = a * a + b * b;
    const double d1 = (x1 * a + y1 * b) + c;
    const double n1 = d1 * d1, n2 = d2 * d2;
    const bool out = (n1 >= den1 * thi) | (n2 >= den2 * thi);
    const bool in = (n1 <= den1 * tlo) & (n2 <= den2 * tlo);
    int r = (in && !out) ? 1 : 0;
    if (!(pos_finite(den1) && pos_finite(den2) && (in || out))) {
        const double s2 = 1. / den2;
        const double s1 = 1. / den1;
        r = (float)fmax(d1 * d1 * s1, d2 * d2 * s2) <= tf ? 1 : 0;
    }
    return r;
}

// ---- single-precision filter ---------------------------------------------------
// Most (model, point) tests are nowhere near the threshold, and for those a
// float32 evaluation with a rigorous error bound gives the same answer as the
// double-precision procedure at a fraction of the cost (packed f32 FMAs, two
// points per instruction).  Only the evaluations the bound cannot settle are
// redone exactly.
//
// Bound.  u = 2^-24.  Every coordinate has |x| <= R.  For a row (p,q,r) of the
// model, a = p*x + q*y + r evaluated as fma(p,x,fma(q,y,r)) on float-rounded
// inputs differs from the double-precision value by at most
// da = 8u*((|p|+|q|)R + |r|) =: 8u*A, and the residual d = x'*a + y'*b + c by at
// most dd = 16u*(R(A+B)+C).  With 2|v|e <= eps*v^2 + e^2/eps,
//     d*^2   in  d^2 (1 -+ eps) -+ Kn,   Kn = dd^2 (1 + 1/eps)
//     den*   in  den (1 -+ eps) -+ Kd,   Kd = (da^2 + db^2 [+ ...]) (1 + 1/eps)
// (starred = double-precision values), so
//     d^2 (1+eps) + Kn <= (den (1-eps) - Kd) T-   proves  d*^2/den* <= T-
//     d^2 (1-eps) - Kn >= (den (1+eps) + Kd) T+   proves  d*^2/den* >= T+
// with T-+ = tf (1 -+ 2^-16), inside which the double-precision procedure
// (2^-20 margins, then a float-rounded quotient) cannot decide otherwise.  The
// constants carry 2^-18 of extra slack for the roundings of the test itself.
typedef float f2v __attribute__((ext_vector_type(2)));

struct FilterConst {
    float c1, c2, c3, c4;  // multipliers of d^2 and den in the two tests
    float kin[2], kout[2]; // additive terms ([0] is used; [1] mirrors it)
};

__device__ __forceinline__ float f_up(double v) { return (float)(v * (1.0 + 1.0 / 1048576.0)); }
__device__ __forceinline__ float f_down(double v) { return (float)(v * (1.0 - 1.0 / 1048576.0)); }

template <int MODEL>
__device__ __forceinline__ FilterConst make_filter(const double* M, double R, float tf)
{
    const double u = 1.0 / 16777216.0, eps = 1.0 / 512.0, g = 1.0 + 1.0 / 262144.0;
    const double Rr = R * (1.0 + 1.0 / 1048576.0);
    double m[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) m[i] = fabs(M[i]);
    const double Tin = (double)tf * (1.0 - 1.0 / 65536.0), Tout = (double)tf * (1.0 + 1.0 / 65536.0);
    const double ke = (1.0 + 1.0 / eps) * (1.0 + 1.0 / 1024.0);
    double kn[2], kd[2], smax;
    // rows of the model against (x1,y1,1): A_i = (|m[3i]|+|m[3i+1]|) R + |m[3i+2]|
    const double r0 = (m[0] + m[1]) * Rr + m[2], r1 = (m[3] + m[4]) * Rr + m[5], r2 = (m[6] + m[7]) * Rr + m[8];
    // columns against (x2,y2,1)
    const double q0 = (m[0] + m[3]) * Rr + m[6], q1 = (m[1] + m[4]) * Rr + m[7], q2 = (m[2] + m[5]) * Rr + m[8];
    if (MODEL == SFM_GPU_MODEL_FUNDAMENTAL) {
        // The two residuals x2^T F x1 and x1^T F^T x2 are one real number; their
        // double-precision values differ from it by rounding only (2^-50 of the
        // term sums), so the filter evaluates it once and tests it against the
        // smaller of the two denominators with the larger of the two slacks.
        const double s2 = Rr * (r0 + r1) + r2;  // term sum of x2^T (F x1)
        const double s1 = Rr * (q0 + q1) + q2;  // term sum of x1^T (F^T x2)
        const double sm = fmax(s1, s2) * (1.0 + 1.0 / 1048576.0);
        kn[0] = (16 * u * sm) * (16 * u * sm) * ke;
        kd[0] = fmax((8 * u * q0) * (8 * u * q0) + (8 * u * q1) * (8 * u * q1),
                     (8 * u * r0) * (8 * u * r0) + (8 * u * r1) * (8 * u * r1)) * ke;
        kn[1] = kn[0];
        kd[1] = kd[0];
        smax = fmax(sm, fmax(fmax(r0, r1), fmax(q0, q1)));
    } else {
        const double ss = Rr * (r0 + r1) + r2;
        kn[0] = (16 * u * ss) * (16 * u * ss) * ke;
        kd[0] = ((8 * u * r0) * (8 * u * r0) + (8 * u * r1) * (8 * u * r1) + (8 * u * q0) * (8 * u * q0) +
                 (8 * u * q1) * (8 * u * q1)) * ke;
        kn[1] = kn[0];
        kd[1] = kd[0];
        smax = fmax(fmax(ss, fmax(q0, q1)), fmax(r0, r1));
    }
    FilterConst f;
    f.c1 = f_up((1.0 + eps) * g);
    f.c2 = f_down((1.0 - eps) * Tin / g);
    f.c3 = f_down((1.0 - eps) / g);
    f.c4 = f_up((1.0 + eps) * Tout * g);
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        f.kin[k] = f_up(fmax((kn[k] + kd[k] * Tin) * g, 1e-30));
        f.kout[k] = f_up(fmax((kn[k] + kd[k] * Tout) * g, 1e-30));
    }
    // anything that could overflow or is not a number: no test can pass
    const bool ok = (smax * smax * fmax(1.0, (double)f.c4) * 8.0 < 1e37) && (R < 1e18) && (tf > 1e-12f) &&
                    (tf < 1e12f);
    if (!ok) {
        f.kin[0] = f.kin[1] = f.kout[0] = f.kout[1] = __builtin_inff();
    }
    return f;
}

__device__ __forceinline__ f2v fma2(f2v a, f2v b, f2v c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ f2v splat(float v) { return (f2v)(v); }
__device__ __forceinline__ f2v max2(f2v a, f2v b)
{
    f2v r;
    r.x = fmaxf(a.x, b.x);
    r.y = fmaxf(a.y, b.y);
    return r;
}

// Two points against one model: squared residual and the denominator to test
// it against (the smaller of the two sides for the symmetric epipolar distance).
struct PairTerms {
    f2v n, den;
};
__device__ __forceinline__ f2v min2(f2v a, f2v b)
{
    f2v r;
    r.x = fminf(a.x, b.x);
    r.y = fminf(a.y, b.y);
    return r;
}
template <int MODEL>
__device__ __forceinline__ PairTerms filter_terms(const float* Mf, f2v x1, f2v y1, f2v x2, f2v y2)
{
    PairTerms t;
    const f2v a = fma2(splat(Mf[0]), x1, fma2(splat(Mf[1]), y1, splat(Mf[2])));
    const f2v b = fma2(splat(Mf[3]), x1, fma2(splat(Mf[4]), y1, splat(Mf[5])));
    const f2v c = fma2(splat(Mf[6]), x1, fma2(splat(Mf[7]), y1, splat(Mf[8])));
    const f2v d = fma2(x2, a, fma2(y2, b, c));
    const f2v at = fma2(splat(Mf[0]), x2, fma2(splat(Mf[3]), y2, splat(Mf[6])));
    const f2v bt = fma2(splat(Mf[1]), x2, fma2(splat(Mf[4]), y2, splat(Mf[7])));
    t.n = d * d;
    if (MODEL == SFM_GPU_MODEL_FUNDAMENTAL) t.den = min2(fma2(a, a, b * b), fma2(at, at, bt * bt));
    else t.den = fma2(a, a, fma2(b, b, fma2(at, at, bt * bt)));
    return t;
}
// h_in <= 0 proves an inlier, h_out >= 0 an outlier
__device__ __forceinline__ f2v filter_h_in(const PairTerms& t, const FilterConst& fc)
{
    return fma2(t.n, splat(fc.c1), fma2(t.den, splat(-fc.c2), splat(fc.kin[0])));
}
__device__ __forceinline__ f2v filter_h_out(const PairTerms& t, const FilterConst& fc)
{
    return fma2(t.n, splat(fc.c3), fma2(t.den, splat(-fc.c4), splat(-fc.kout[0])));
}

// One lane's model during scoring: float32 copies of its coefficients and the
// constants of the float32 filter.  The double-precision coefficients stay in
// memory and are fetched again for the few tests that need them, which keeps the
// scoring loops at a register count that lets six or more wavefronts share a
// SIMD.
template <int MODEL>
struct ScoreLane {
    const double* Mg;
    float Mf[9];
    FilterConst fc;
    __device__ __forceinline__ void load(const RansacProb& pb, int e, float R)
    {
        Mg = pb.models + (size_t)(e / MAX_MODELS) * 90 + (e % MAX_MODELS) * 9;
        double M[9];
#pragma unroll
        for (int i = 0; i < 9; ++i) M[i] = Mg[i];
        fc = make_filter<MODEL>(M, (double)R, pb.tf);
        // (a model the filter refuses -- infinite slacks -- gets zero coefficients: every
        // filter value of it is then finite, +inf or -inf, never a NaN)
        const bool refused = fc.kin[0] == __builtin_inff();
#pragma unroll
        for (int k = 0; k < 9; ++k) Mf[k] = refused ? 0.f : (float)M[k];
    }
};

constexpr int SCORE_QCAP = 40;  // tests a lane can park per span before it falls back to exact evaluation on the spot

// Inliers of the lane's model among the points [i0, i1) of the problem (i0 a
// multiple of 4).  The points are the same for every lane and arrive by scalar
// loads (nothing writes them while a scoring kernel runs), so a point costs each
// model nothing but its own arithmetic.  Tests the filter cannot settle are
// parked per lane (`parked`: SCORE_QCAP x 64 offsets of this wavefront in LDS)
// and redone in double precision after the span, so that one lane's doubt does
// not make the other 63 wait at every such point.
template <int MODEL>
__device__ __forceinline__ int score_span(const RansacProb& pb, const ScoreLane<MODEL>& m, int i0, int i1,
                                          unsigned short* parked, int lane, int use_filter,
                                          unsigned long long* t_after_filter = nullptr)
{
    typedef const double __attribute__((address_space(4))) cdouble;
    typedef const f2v __attribute__((address_space(4))) cf2v;
    typedef const float __attribute__((address_space(4))) cfloat;
    const float tf = pb.tf;
    const double tlo = pb.tlo, thi = pb.thi;
    cdouble* X1 = (cdouble*)(uintptr_t)pb.x1;
    cdouble* X2 = (cdouble*)(uintptr_t)pb.x2;
    cfloat* S = (cfloat*)(uintptr_t)pb.soa;
    auto exact = [&](const double* M, int i) -> int {
        if (MODEL == SFM_GPU_MODEL_ESSENTIAL)
            return sampson_inlier_u(M, X1[2 * i], X1[2 * i + 1], X2[2 * i], X2[2 * i + 1], tf, tlo, thi);
        return symepi_inlier_u(M, X1[2 * i], X1[2 * i + 1], X2[2 * i], X2[2 * i + 1], tf, tlo, thi);
    };
    int cnt = 0;
    int i = i0;
    int n_parked = 0;
    if (use_filter) {
        auto park = [&](int idx) {
#ifdef SFM_GPU_TIMING_BUILD
            if (use_filter >= 2) return;  // timing experiments: unsettled tests are dropped (wrong counts)
#endif
            if (n_parked < SCORE_QCAP) {
                parked[n_parked * 64 + lane] = (unsigned short)(idx - i0);
                ++n_parked;
            } else {
                // rare: the lane's queue is full.  (The pointer is made opaque so that these loads
                // stay here: hoisted out of the loop they would hold 18 registers through it.)
                const double* mp = m.Mg;
                asm volatile("" : "+v"(mp));
                double M[9];
#pragma unroll
                for (int k = 0; k < 9; ++k) M[k] = mp[k];
                cnt += exact(M, idx);
            }
        };
        // four points per step (one 64-byte block: X1[4] Y1[4] X2[4] Y2[4]), the
        // next four already on their way
        const int i4 = i0 + ((i1 - i0) & ~3);
        f2v nx[8];
        auto fetch = [&](int at) {
            cf2v* B = (cf2v*)(S + (size_t)(at >> 2) * 16);
#pragma unroll
            for (int k = 0; k < 8; ++k) nx[k] = B[k];
        };
        if (i < i4) fetch(i);
        for (; i < i4; i += 4) {
            f2v cur[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) cur[k] = nx[k];
            fetch(i + 4 < i4 ? i + 4 : i);  // the last step re-reads its own points
            const PairTerms tA = filter_terms<MODEL>(m.Mf, cur[0], cur[2], cur[4], cur[6]);
            const PairTerms tB = filter_terms<MODEL>(m.Mf, cur[1], cur[3], cur[5], cur[7]);
            const f2v hinA = filter_h_in(tA, m.fc), hinB = filter_h_in(tB, m.fc);
            const f2v houtA = filter_h_out(tA, m.fc), houtB = filter_h_out(tB, m.fc);
            // Verdicts from the sign bits, without compares or lane masks (the loop is
            // bound by the instructions a wavefront issues): h_in < 0 proves an inlier,
            // h_out < 0 is an outlier not proven.  Zeros read as "not proven" on either
            // side, which only sends the test to the double-precision pass, and no value
            // here is a NaN: the coordinates are finite (the caller checks the problem's
            // largest one), and a model whose terms could overflow has its float32
            // coefficients set to zero and infinite slacks (ScoreLane::load).
            // bit 3 = point i ... bit 0 = point i + 3
            unsigned sin = __float_as_uint(hinA.x) >> 31;
            sin = __builtin_amdgcn_alignbit(sin, __float_as_uint(hinA.y), 31);
            sin = __builtin_amdgcn_alignbit(sin, __float_as_uint(hinB.x), 31);
            sin = __builtin_amdgcn_alignbit(sin, __float_as_uint(hinB.y), 31);
            cnt += __builtin_popcount(sin);
            unsigned sout = __float_as_uint(houtA.x) >> 31;
            sout = __builtin_amdgcn_alignbit(sout, __float_as_uint(houtA.y), 31);
            sout = __builtin_amdgcn_alignbit(sout, __float_as_uint(houtB.x), 31);
            sout = __builtin_amdgcn_alignbit(sout, __float_as_uint(houtB.y), 31);
            unsigned doubt = sout & ~sin;
            while (__builtin_amdgcn_ballot_w64(doubt != 0u) != 0ull) {  // in most steps some lane has one, hardly ever two
                if (doubt != 0u) {
                    const int b = 31 - __builtin_clz(doubt);
                    park(i + 3 - b);
                    doubt &= ~(1u << b);
                }
            }
        }
    }
#ifdef SFM_GPU_TIMING_BUILD
    if (t_after_filter) *t_after_filter = __builtin_amdgcn_s_memrealtime();
#endif
    // whatever is left needs the double-precision coefficients
    if (__builtin_amdgcn_ballot_w64(i < i1 || n_parked > 0) != 0ull) {
        double M[9];
#pragma unroll
        for (int k = 0; k < 9; ++k) M[k] = m.Mg[k];
        for (; i < i1; ++i) cnt += exact(M, i);
        for (int k = 0; k < n_parked; ++k) cnt += exact(M, i0 + (int)parked[k * 64 + lane]);
    }
    return cnt;
}

// Inlier counts: one thread per listed candidate model, which it keeps in
// registers (score_span).
// grid = (groups of 64 models, point chunks from chunk0 on, problems); a
// workgroup strides over further groups / chunks when the grid is smaller than
// the work, keeping its models in registers from chunk to chunk.
#ifndef SFM_SCORE_WAVES
#define SFM_SCORE_WAVES 1
#endif
constexpr int SCORE_WAVES = SFM_SCORE_WAVES;  // independent wavefronts per workgroup
// The points a launch covers are [result[slot_begin], min(n, result[slot_end])) of
// its problem (a slot < 0 stands for 0 / n), the models the first
// min(R_NLIST, list_limit) of the list.
// live_slot >= 0 (not with STRIDED): the launch is a stage over a list put in
// order by ransac_sortplan_kernel.  A wavefront whose models have all fallen out
// of reach of the bound R_BOUND (count so far + points left < bound: they can
// neither win nor tie) stops at once; result[live_slot] bounds the groups that
// can hold anything else, and the wavefronts that go on record the same for the
// next stage in result[live_slot + 1].  The pieces of a stretch are then at
// least SCORE_MIN_PIECE points long.
// (measured on the bench, ms per step: pieces of 64 / 128 / 256 / 512 points 0.861 / 0.852 / 0.873 / 0.923;
// 8192 / 16384 / 32768 items 0.887 / 0.852 / 0.852)
constexpr int SCORE_MIN_PIECE = 128;
constexpr int SCORE_STAGE_ITEMS = 16384;  // wavefronts a stage launch may use over the batch when the list is short
// Timing build only: stamps of the scoring wavefronts that reach their end, four 64-bit words each
// (scripts/experiments/score_residency.py): realtime counter (100 MHz) at entry; at exit; hardware id (16 bits) |
// XCC id << 16 | ticks from entry to "models known to be live" << 32 | to "model and filter set up" << 48;
// launch tag (32 bits) | ticks from entry to the end of the float32 loop << 32 | shader cycles of its life / 16 << 48.
#ifdef SFM_GPU_TIMING_BUILD
constexpr unsigned SCORE_STAMPS = 1u << 18;
__device__ unsigned long long g_score_stamps[SCORE_STAMPS * 4];
__device__ unsigned g_score_stamp_count[64];  // slot = 64 * count[b] + b, b = workgroup % 64: no queue on one counter
#define SFM_SCORE_T0()                                                       \
    const unsigned long long stamp_t0 = __builtin_amdgcn_s_memrealtime();   \
    const unsigned long long stamp_c0 = __builtin_amdgcn_s_memtime();       \
    unsigned long long stamp_tA = stamp_t0, stamp_tB = stamp_t0, stamp_tC = stamp_t0
#define SFM_SCORE_MARK(v) v = __builtin_amdgcn_s_memrealtime()
#define SFM_SCORE_TC &stamp_tC
#define SFM_SCORE_STAMP(tag)                                                                                   \
    do {                                                                                                       \
        if ((threadIdx.x & 63) == 0) {                                                                         \
            const unsigned lin_ = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);              \
            const unsigned slot_ = 64u * atomicAdd(&g_score_stamp_count[lin_ & 63u], 1u) + (lin_ & 63u);       \
            if (slot_ < SCORE_STAMPS) {                                                                        \
                unsigned hw_, xcc_;                                                                            \
                asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw_));                              \
                asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc_));                            \
                g_score_stamps[4 * slot_] = stamp_t0;                                                          \
                g_score_stamps[4 * slot_ + 1] = __builtin_amdgcn_s_memrealtime();                              \
                g_score_stamps[4 * slot_ + 2] = (unsigned long long)(hw_ & 0xFFFFu) |                          \
                                                ((unsigned long long)(xcc_ & 0xFu) << 16) |                    \
                                                (((stamp_tA - stamp_t0) & 0xFFFFull) << 32) |                  \
                                                (((stamp_tB - stamp_t0) & 0xFFFFull) << 48);                   \
                g_score_stamps[4 * slot_ + 3] =                                                                \
                    (unsigned long long)(unsigned)(tag) | (((stamp_tC - stamp_t0) & 0xFFFFull) << 32) |        \
                    ((((__builtin_amdgcn_s_memtime() - stamp_c0) >> 4) & 0xFFFFull) << 48);                    \
            }                                                                                                  \
        }                                                                                                      \
    } while (0)
#else
#define SFM_SCORE_T0() do { } while (0)
#define SFM_SCORE_MARK(v) do { } while (0)
#define SFM_SCORE_TC nullptr
#define SFM_SCORE_STAMP(tag) do { } while (0)
#endif
template <int MODEL, bool STRIDED>
__global__ void __launch_bounds__(SCORE_WAVES * 64) ransac_score_kernel(const RansacProb* __restrict__ probs, int chunk_pts_arg,
                                                          int use_filter, int slot_begin, int slot_end,
                                                          int list_limit, int live_slot)
{
    // live_slot == -2 (the opening stretch): grid = (problems, pieces, groups).  Consecutive
    // workgroups are placed on the same compute unit until it is full, so problems vary fastest
    // and a problem's groups are spread over the whole chip.
    // live_slot >= 0 (a stage): grid = (problems, work items).  Only the result[live_slot] groups
    // at the head of the ordered list can have work, so the problem's items go to them alone:
    // item -> (group, piece), pieces as short as SCORE_MIN_PIECE points when the items allow it,
    // group-major, so that the wavefronts with work are dispatched first and side by side in the
    // grid only across problems.
    SFM_SCORE_T0();
    const bool stage = !STRIDED && live_slot >= 0;
    const bool by_group = !STRIDED && live_slot == -2;
    unsigned bx = by_group ? blockIdx.z : blockIdx.x, by = blockIdx.y;
    const unsigned bz = (by_group || stage) ? blockIdx.x : blockIdx.z;
    const unsigned gdx = by_group ? gridDim.z : gridDim.x;
    const RansacProb pb = probs[bz];
    // The problem's whole result block in one scalar load (nothing this launch reads from it is
    // written during the launch: a stage writes the live mark of the NEXT stage only), so that
    // what the wavefront needs to know before its first model comes in one round trip, not one
    // per early-exit test.
    typedef const int __attribute__((address_space(4))) cint;
    int res[R_COUNT];
    {
        cint* RES = (cint*)(uintptr_t)pb.result;
#pragma unroll
        for (int k = 0; k < R_COUNT; ++k) res[k] = RES[k];
    }
    const int n_all = *(cint*)(uintptr_t)pb.n_ptr;
    const int nlist = min(res[R_NLIST], list_limit);
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int pts_begin = slot_begin >= 0 ? res[slot_begin] : 0;
    int n = n_all;
    if (slot_end >= 0) n = min(n, res[slot_end]);
    int chunk_pts;
    if (stage) {
        const int live = res[live_slot];
        const int len = n - pts_begin;
        if (live <= 0 || len <= 0) return;
        // (the grid has at least as many items as the list has groups, hence >= live)
        const int most = (len + SCORE_MIN_PIECE - 1) / SCORE_MIN_PIECE;
        const int pieces = max(1, min(most, (int)gridDim.y * SCORE_WAVES / live));
        chunk_pts = ((len + pieces - 1) / pieces + 3) & ~3;
        const unsigned item = blockIdx.y * SCORE_WAVES + wv;
        bx = item / (unsigned)pieces;
        by = item - bx * (unsigned)pieces;
        if ((int)bx >= live) return;
        if ((int)bx * 64 >= nlist) return;
    } else {
        // chunk_pts_arg <= 0: the problem's stretch in equal pieces, one per grid row
        chunk_pts = chunk_pts_arg > 0 ? chunk_pts_arg
                                      : (((max(n - pts_begin, 0) + (int)gridDim.y - 1) / (int)gridDim.y + 3) & ~3);
    }
    if (!stage && ((int)bx * SCORE_WAVES + wv) * 64 >= nlist) return;
    if (pts_begin + (int)by * chunk_pts >= n) return;  // stage boundaries and chunk_pts are multiples of 4
    const int lane = threadIdx.x & 63;
    __shared__ unsigned short parked_all[SCORE_WAVES * SCORE_QCAP * 64];
    unsigned short* parked = parked_all + wv * (SCORE_QCAP * 64);
    const float R = __int_as_float(res[R_MAXABS]);
    if (!(R < 1e18f)) use_filter = 0;  // coordinates that are huge, infinite or not numbers: double precision only
    for (int g0 = (stage ? bx : bx * SCORE_WAVES + wv) * 64; g0 < nlist; g0 += gdx * SCORE_WAVES * 64) {
        const bool active = g0 + lane < nlist;
        // the stages read the ordered list where ransac_sortplan_kernel left it
        const int32_t* lst = stage ? pb.list2 + res[R_TOP] : pb.list;
        const int e = lst[active ? g0 + lane : g0];
        if (stage) {
            // (every piece of the group sees the same counts: the stage before is complete)
            const int bound = res[R_BOUND];
            const bool alive = active && pb.counts[e] + (n_all - pts_begin) >= bound;
            if (__builtin_amdgcn_ballot_w64(alive) == 0ull) return;
            if (by == 0 && lane == 0) atomicMax(&pb.result[live_slot + 1], g0 / 64 + 1);
        }
        SFM_SCORE_MARK(stamp_tA);
        ScoreLane<MODEL> m;
        m.load(pb, e, R);
        SFM_SCORE_MARK(stamp_tB);
        int cnt = 0;
        for (int i0 = pts_begin + by * chunk_pts; i0 < n; i0 += gridDim.y * chunk_pts) {
            cnt += score_span<MODEL>(pb, m, i0, min(n, i0 + chunk_pts), parked, lane, use_filter, SFM_SCORE_TC);
            if (!STRIDED) break;  // the grid covers every chunk: straight-line code
        }
        if (active && cnt != 0) atomicAdd(&pb.counts[e], cnt);
        if (!STRIDED) break;
    }
    SFM_SCORE_STAMP(live_slot);
}

// Between the opening stretch [0, R_S0), on which every listed model has been
// counted, and the stages that follow.  One workgroup per problem:
//  * orders the list by descending partial count (counting sort; the order
//    inside a bucket is whatever the atomics make it: only cost depends on it),
//  * completes the SORT_TOP best of them over all remaining points and takes
//    them off the list: the largest of their counts is the bound B (R_BOUND),
//    a lower bound of the winner's count,
//  * sets the ends R_S1, R_MID, R_S3 of the stages that follow and the count of
//    groups the first of them has to look at (R_LIVE).
// (measured on the bench, ms per step: 1 / 2 / 4 / 8 / 16 models completed 0.835 / 0.830 / 0.833 / 0.8375 / 0.849:
// few models finish sooner, many raise the bound a little)
#ifndef SFM_SORT_TOP
#define SFM_SORT_TOP 2
#endif
constexpr int SORT_TOP = SFM_SORT_TOP;  // a power of two, at most 16

// Entries base + tid + 1024 k (k < 8) of a problem's list and their counts, all
// loads of one kind issued together: these one-workgroup kernels are chains of
// dependent memory round trips, and a loop of load -> load -> use pays two of
// them per turn.
__device__ __forceinline__ void load_entries8(const RansacProb& pb, int base, int nlist, int tid, int (&e)[8],
                                              int (&c)[8])
{
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int p = base + tid + 1024 * k;
        e[k] = p < nlist ? pb.list[p] : -1;
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) c[k] = e[k] >= 0 ? pb.counts[e[k]] : 0;
}

// Inliers of model M among the points first, first + stride, ... < n, by the
// procedure of the mask, eight points' loads in flight at a time.
__device__ __forceinline__ int count_exact_strided(const RansacProb& pb, const double* M, int first, int stride, int n)
{
    int add = 0;
    for (int i = first; i < n; i += 8 * stride) {
        if (pb.model == SFM_GPU_MODEL_ESSENTIAL) {
            double2 a[8], b[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int ii = min(i + k * stride, n - 1);
                a[k] = *(const double2*)(pb.x1 + 2 * ii);
                b[k] = *(const double2*)(pb.x2 + 2 * ii);
            }
#pragma unroll
            for (int k = 0; k < 8; ++k)
                if (i + k * stride < n) add += sampson_inlier(M, a[k].x, a[k].y, b[k].x, b[k].y, pb.tf, pb.tlo, pb.thi);
        } else {
            float2 a[8], b[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int ii = min(i + k * stride, n - 1);
                a[k] = *(const float2*)(pb.f1 + 2 * ii);
                b[k] = *(const float2*)(pb.f2 + 2 * ii);
            }
#pragma unroll
            for (int k = 0; k < 8; ++k)
                if (i + k * stride < n) add += symepi_inlier(M, a[k].x, a[k].y, b[k].x, b[k].y, pb.tf, pb.tlo, pb.thi);
        }
    }
    return add;
}

__global__ void __launch_bounds__(1024) ransac_sortplan_kernel(const RansacProb* __restrict__ probs)
{
    __shared__ int hist[SCORE_OPENING + 1];
    __shared__ int s_add[SORT_TOP], s_top_e[SORT_TOP], s_top_c[SORT_TOP];
    const RansacProb pb = probs[blockIdx.x];
    const int tid = threadIdx.x;
    const int n = *pb.n_ptr;
    const int nlist = pb.result[R_NLIST];
    const int bound0 = pb.result[R_BOUND];
    const int s0 = min(n, SCORE_OPENING);
    if (tid == 0) {
        pb.result[R_S0] = s0;
        pb.result[R_S1] = n;
        pb.result[R_MID] = n;
        pb.result[R_S3] = n;
        for (int k = 0; k < 5; ++k) pb.result[R_LIVE + k] = 0;
        pb.result[R_TOP] = 0;
    }
    if (nlist <= 0 || s0 >= n) return;
    for (int b = tid; b <= SCORE_OPENING; b += 1024) hist[b] = 0;
    if (tid < SORT_TOP) s_add[tid] = 0;
    __syncthreads();
    int e[8], c[8];  // the first 8192 entries stay in registers for the second pass
    for (int base = 0; base < nlist; base += 8192) {
        int eb[8], cb[8];
        load_entries8(pb, base, nlist, tid, eb, cb);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            cb[k] = min(max(cb[k], 0), SCORE_OPENING);
            if (eb[k] >= 0) atomicAdd(&hist[cb[k]], 1);
            if (base == 0) {
                e[k] = eb[k];
                c[k] = cb[k];
            }
        }
    }
    __syncthreads();
    // hist[c] := first slot of bucket c, buckets in descending order of c
    if (tid < 64) {
        int running = 0;
        for (int c0 = SCORE_OPENING; c0 >= 0; c0 -= 64) {
            const int cc = c0 - tid;
            const int v = cc >= 0 ? hist[cc] : 0;
            int incl = v;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const int o = __shfl_up(incl, off);
                if (tid >= off) incl += o;
            }
            if (cc >= 0) hist[cc] = running + incl - v;
            running += __shfl(incl, 63);
        }
    }
    __syncthreads();
    // (whoever places an entry among the first SORT_TOP also leaves it in LDS: the
    // completion below then needs no trip to memory to learn which models it is about)
    auto place = [&](int ent, int cnt) {
        const int slot = atomicAdd(&hist[cnt], 1);
        pb.list2[slot] = ent;
        if (slot < SORT_TOP) {
            s_top_e[slot] = ent;
            s_top_c[slot] = cnt;
        }
    };
#pragma unroll
    for (int k = 0; k < 8; ++k)
        if (e[k] >= 0) place(e[k], c[k]);
    for (int base = 8192; base < nlist; base += 8192) {
        int eb[8], cb[8];
        load_entries8(pb, base, nlist, tid, eb, cb);
#pragma unroll
        for (int k = 0; k < 8; ++k)
            if (eb[k] >= 0) place(eb[k], min(max(cb[k], 0), SCORE_OPENING));
    }
    __syncthreads();
    // the best SORT_TOP, 1024 / SORT_TOP threads each, over the points the opening has not seen
    const int top = min(SORT_TOP, nlist);
    {
        constexpr int TPM = 1024 / SORT_TOP;  // threads per model
        const int t = tid / TPM, j = tid % TPM;
        int add = 0;
        if (t < top) {
            const int et = s_top_e[t];
            const double* Mg = pb.models + (size_t)(et / MAX_MODELS) * 90 + (et % MAX_MODELS) * 9;
            double M[9];
#pragma unroll
            for (int i = 0; i < 9; ++i) M[i] = Mg[i];
            add = count_exact_strided(pb, M, s0 + j, TPM, n);
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) add += __shfl_xor(add, off);
        if ((tid & 63) == 0 && t < top && add) atomicAdd(&s_add[t], add);
    }
    __syncthreads();
    if (tid < 64) {
        int full = 0;
        if (tid < top) {
            full = s_top_c[tid] + s_add[tid];  // (the opening count of a listed model is its bucket: never clamped)
            pb.counts[s_top_e[tid]] = full;
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) full = max(full, __shfl_xor(full, off));
        if (tid == 0) {
            const int bound = max(bound0, full);
            pb.result[R_BOUND] = bound;
            pb.result[R_NLIST] = nlist - top;
            pb.result[R_TOP] = top;
            // The list is in descending order of the opening counts and hist[c] now is
            // the number of models with at least c: the models the first stage cannot
            // rule out at the door (count + points left >= bound) are a prefix of it.
            const int need = bound - (n - s0);
            const int reach = need <= 0 ? nlist : (need > SCORE_OPENING ? 0 : hist[need]);
            pb.result[R_LIVE] = (max(reach - top, 0) + 63) / 64;
            auto up128 = [](int v) { return (v + 127) / 128 * 128; };
            {
                // A model that misses nearly every point falls out of reach once its misses
                // exceed n - bound: the first stage ends just past that, where most of the
                // list stops; the later stages grow, as fewer and fewer models reach them.
                // (three stages instead of four: 0.831-0.836 against 0.823 ms per step; other
                // shares for the second and third stage, 1/10 + 1/4 or 1/6 + 1/2: no difference)
                const int b1 = max(up128((int)(1.06 * (double)(n - bound)) + 16), s0 + 128);
                const int b2 = b1 + up128(max(256, (n - b1) / 8));
                const int b3 = b2 + up128(max(640, (n - b2) / 3));
                pb.result[R_S1] = n - b1 < 256 ? n : b1;
                pb.result[R_MID] = n - b2 < 256 ? n : b2;
                pb.result[R_S3] = n - b3 < 256 ? n : b3;
            }
        }
    }
}

// Staged scoring, after every listed model has been scored on the opening stretch
// [0, s0) of the points: the best partial count there estimates the inlier share
// (lowered by two standard deviations of a count of s0 trials, since the best of
// thousands of noisy counts flatters it), hence the outliers to expect.  The first
// stage has to outnumber the outliers for the bound to drop anything: it is
// extended to 1.3 x that estimate (at least 15 % of the points), in pieces of 128,
// at most `cover` points (what the next launch's grid reaches); the second cut comes 35 % into the rest when
// enough is left to pay for it.  One workgroup per problem.  Only the cost
// depends on these choices, never the result.
__global__ void __launch_bounds__(1024) ransac_plan_kernel(const RansacProb* __restrict__ probs, int cover,
                                                           int one_cut)
{
    __shared__ int s_best;
    const RansacProb pb = probs[blockIdx.x];
    const int tid = threadIdx.x;
    const int n = *pb.n_ptr;
    const int nlist = pb.result[R_NLIST];
    if (tid == 0) s_best = 0;
    __syncthreads();
    int best = 0;
    for (int p = tid; p < nlist; p += 1024) best = max(best, pb.counts[pb.list[p]]);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) best = max(best, __shfl_xor(best, off));
    if ((tid & 63) == 0) atomicMax(&s_best, best);
    __syncthreads();
    if (tid == 0) {
        const int s0 = min(n, SCORE_OPENING);
        int s1 = n, mid = n;
        if (s0 > 0 && n > s0) {
            const double fh = ((double)s_best + 1.0) / ((double)s0 + 2.0);
            const double sd = sqrt(fmax(fh * (1.0 - fh), 0.0) / (double)s0);
            const double ft = fmax(fh - 2.0 * sd, 0.0);
            const double outl = (double)n * (1.0 - ft);
            // at least 15 % of the points: with few outliers the cut is cheap to make a little
            // later, and it then also drops the half-good models (samples with one wrong match)
            s1 = ((int)fmax(1.3 * outl, 0.15 * (double)n) + 1 + 127) / 128 * 128;
            s1 = max(s1, s0);
            s1 = min(s1, max(cover, s0));
            if (n - s1 < 128) s1 = n;  // not worth another launch: everything in the first stage
            if (!one_cut && n - s1 >= 1024) mid = (s1 + (int)(0.35 * (double)(n - s1)) + 127) / 128 * 128;
        }
        pb.result[R_S1] = s1;
        pb.result[R_MID] = mid;
    }
}

// Between two scoring stages of a fixed-count batch (points [0, pts_done) are
// counted for every listed model): the listed model with the largest partial
// count is completed here over the remaining points, which gives a lower bound
// B on the count the winner will have; a model whose partial count plus all the
// remaining points stays below B can neither win nor tie, so it leaves the
// list.  Survivors keep their scan order.  One workgroup per problem.
__global__ void __launch_bounds__(1024) ransac_prune_kernel(const RansacProb* __restrict__ probs, int slot_done)
{
    __shared__ unsigned long long sh[16];
    __shared__ int wsum[16];
    __shared__ int s_running, s_bound, s_pos;
    const RansacProb pb = probs[blockIdx.x];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int n = *pb.n_ptr;
    const int nlist = pb.result[R_NLIST];
    const int pts_done = pb.result[slot_done];
    const int bound0 = pb.result[R_BOUND];
    if (pts_done >= n || nlist <= 0) return;
    // the listed model with the largest partial count (the first 8192 entries and
    // their counts stay in registers for the compaction)
    int e[8], c[8];
    unsigned long long best = 0;
    for (int base = 0; base < nlist; base += 8192) {
        int eb[8], cb[8];
        load_entries8(pb, base, nlist, tid, eb, cb);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            if (eb[k] >= 0) {
                const unsigned long long key =
                    ((unsigned long long)(unsigned)cb[k] << 32) | (0xFFFFFFFFu - (unsigned)(base + tid + 1024 * k));
                best = key > best ? key : best;
            }
            if (base == 0) {
                e[k] = eb[k];
                c[k] = cb[k];
            }
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const unsigned long long o = __shfl_xor(best, off);
        best = o > best ? o : best;
    }
    if (lane == 0) sh[wv] = best;
    __syncthreads();
    best = sh[0];
    for (int w = 1; w < 16; ++w) best = sh[w] > best ? sh[w] : best;
    const int pos = (int)(0xFFFFFFFFu - (unsigned)best);
    const int ebest = pb.list[pos];
    // its count over the remaining points, by the procedure of the mask
    {
        const double* Mg = pb.models + (size_t)(ebest / MAX_MODELS) * 90 + (ebest % MAX_MODELS) * 9;
        double M[9];
#pragma unroll
        for (int i = 0; i < 9; ++i) M[i] = Mg[i];
        int add = count_exact_strided(pb, M, pts_done + tid, 1024, n);
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) add += __shfl_xor(add, off);
        if (lane == 0) wsum[wv] = add;
        __syncthreads();
        if (tid == 0) {
            int tot = 0;
            for (int w = 0; w < 16; ++w) tot += wsum[w];
            const int full = (int)(best >> 32) + tot;
            pb.counts[ebest] = full;
            const int b = max(bound0, full);
            pb.result[R_BOUND] = b;
            s_bound = b;
            s_pos = pos;
            s_running = 0;
        }
        __syncthreads();
    }
    const int bound = s_bound, done_pos = s_pos, rest = n - pts_done;
    // order-preserving compaction in place: a tile is read before anything is
    // written, and writes never reach beyond the tile just read
    for (int p0 = 0; p0 < nlist; p0 += 1024) {
        const int p = p0 + tid;
        int ent = -1;
        if (p < nlist && p != done_pos) {
            int le, lc;
            if (p0 < 8192) {  // tile p0 / 1024 of the registers (the loop is unrolled over the first eight tiles)
                le = e[0];
                lc = c[0];
#pragma unroll
                for (int k = 1; k < 8; ++k)
                    if (p0 == 1024 * k) {
                        le = e[k];
                        lc = c[k];
                    }
            } else {
                le = pb.list[p];
                lc = pb.counts[le];
            }
            if (lc + rest >= bound) ent = le;
        }
        const int keep = ent >= 0 ? 1 : 0;
        int incl = keep;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const int o = __shfl_up(incl, off);
            if (lane >= off) incl += o;
        }
        if (lane == 63) wsum[wv] = incl;
        __syncthreads();
        int base = s_running;
        for (int w = 0; w < wv; ++w) base += wsum[w];
        if (keep) pb.list[base + incl - 1] = ent;
        __syncthreads();
        if (tid == 1023) s_running = base + incl;
        __syncthreads();
    }
    if (tid == 0) pb.result[R_NLIST] = s_running;
}

// Fixed-count winner: the first (hypothesis, model) in scan order whose count is
// the maximum, provided it reaches m — what the sequential loop
// `if (good > max(best, m-1))` ends with.
// The same workgroup then writes the winner's inlier mask and a copy of the
// model (what ransac_mask_kernel does for the adaptive mode).
// With `pairs` (the pair pipeline) it also gathers the inlier matches
// (main.cpp:225-231) and hands the pair's two counts to the host, which saves a
// launch of its own for that.
__global__ void __launch_bounds__(1024) ransac_select_kernel(const RansacProb* __restrict__ probs, int n_hyp,
                                                             const PairDev* __restrict__ pairs)
{
    __shared__ unsigned long long sh[16];
    __shared__ int s_sel[4];  // ok, hypothesis, model, direct
    __shared__ int s_wsum[16];
    const RansacProb pb = probs[blockIdx.x];
    const int n = *pb.n_ptr;
    const int tid = threadIdx.x;
    PairDev pd;
    if (pairs) pd = pairs[blockIdx.x];
    auto report = [&](int n_inl) {  // one thread
        if (!pairs) return;
        *pd.n_inliers = n_inl;
        if (pd.host_counts) {
            pd.host_counts[0] = n;
            pd.host_counts[1] = n_inl;
        }
    };
    if (n < pb.m) {
        if (tid == 0) report(0);
        return;
    }
    const int valid = min(n_hyp, pb.result[R_VALID]);
    unsigned long long best = 0;
    // (ten loads in flight per thread: as a plain loop this scan was a round trip per turn)
    for (int e0 = tid; e0 < valid * 10; e0 += 10 * 1024) {
        int c[10];
#pragma unroll
        for (int k = 0; k < 10; ++k) {
            const int e = e0 + 1024 * k;
            c[k] = e < valid * 10 ? pb.counts[e] : -1;
        }
#pragma unroll
        for (int k = 0; k < 10; ++k) {
            if (c[k] >= pb.m) {
                const unsigned long long key =
                    ((unsigned long long)(unsigned)c[k] << 32) | (0xFFFFFFFFu - (unsigned)(e0 + 1024 * k));
                best = key > best ? key : best;
            }
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const unsigned long long o = __shfl_xor(best, off);
        best = o > best ? o : best;
    }
    if ((tid & 63) == 0) sh[tid >> 6] = best;
    __syncthreads();
    if (tid == 0) {
        for (int i = 1; i < 16; ++i) best = sh[i] > best ? sh[i] : best;
        pb.result[R_ITERS] = valid;
        int ok = 0, hyp = 0, q = 0;
        const int direct = pb.result[R_DIRECT];
        if (direct) {
            ok = pb.nmodels[0] > 0;
            pb.result[R_OK] = ok;
            pb.result[R_HYP] = 0;
            pb.result[R_Q] = 0;
            pb.result[R_NINL] = ok ? n : 0;
        } else if (best != 0) {
            const unsigned e = 0xFFFFFFFFu - (unsigned)best;
            ok = 1;
            hyp = (int)(e / 10);
            q = (int)(e % 10);
            pb.result[R_OK] = 1;
            pb.result[R_HYP] = hyp;
            pb.result[R_Q] = q;
            pb.result[R_NINL] = (int)(best >> 32);
        } else {
            pb.result[R_OK] = 0;
        }
        s_sel[0] = ok;
        s_sel[1] = hyp;
        s_sel[2] = q;
        s_sel[3] = direct;
    }
    __syncthreads();
    if (!s_sel[0]) {  // mask left untouched when no model was found
        if (tid == 0) report(0);
        return;
    }
    const double* Mg = pb.models + (size_t)s_sel[1] * 90 + s_sel[2] * 9;
    double M[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) M[i] = Mg[i];
    if (tid < 9) pb.model_out[tid] = M[tid];
    const bool direct = s_sel[3] != 0;
    const int lane = tid & 63, wv = tid >> 6;
    int running = 0;  // inliers in the tiles before this one (the same in every thread)
    for (int i0 = 0; i0 < n; i0 += 1024) {
        const int i = i0 + tid;
        int f = 0;
        if (i < n) {
            if (direct) {
                f = 1;
            } else if (pb.model == SFM_GPU_MODEL_ESSENTIAL) {
                f = sampson_inlier(M, pb.x1[2 * i], pb.x1[2 * i + 1], pb.x2[2 * i], pb.x2[2 * i + 1], pb.tf, pb.tlo,
                                   pb.thi);
            } else {
                f = symepi_inlier(M, pb.f1[2 * i], pb.f1[2 * i + 1], pb.f2[2 * i], pb.f2[2 * i + 1], pb.tf, pb.tlo,
                                  pb.thi);
            }
            pb.mask[i] = (uint8_t)f;
        }
        if (pairs) {
            // order-preserving compaction of this tile's inlier matches
            const unsigned long long m = __builtin_amdgcn_ballot_w64(f != 0);
            if (lane == 0) s_wsum[wv] = __popcll(m);
            __syncthreads();
            int before = 0, total = 0;
            for (int w = 0; w < 16; ++w) {
                const int c = s_wsum[w];
                if (w < wv) before += c;
                total += c;
            }
            const int slot = running + before + __popcll(m & ((1ull << lane) - 1ull));
            if (f && pd.inliers && slot < pd.capacity) pd.inliers[slot] = pd.good[i];
            running += total;
            __syncthreads();
        }
    }
    if (pairs && tid == 0) report(running);
}

// cv::RANSACUpdateNumIters (OpenCV 4.5.5 calib3d ptsetreg.cpp), as the host's
// update_num_iters spells it.
__device__ int update_num_iters_dev(double p, double ep, int model_points, int max_iters)
{
    p = fmax(p, 0.);
    p = fmin(p, 1.);
    ep = fmax(ep, 0.);
    ep = fmin(ep, 1.);
    double num = fmax(1. - p, 2.2250738585072014e-308);
    double denom = 1. - pow(1. - ep, (double)model_points);
    if (denom < 2.2250738585072014e-308) return 0;
    num = log(num);
    denom = log(denom);
    return denom >= 0 || -num >= max_iters * (-denom) ? max_iters : (int)rint(num / denom);
}

// Adaptive mode.  The loop of RANSACPointSetRegistrator::run over the rows
// [row0, row0 + nrows) whose candidate models have just been counted:
//     for (iter = ...; iter < niters; iter++)
//         for each model of the sample, in solver order:
//             if (good > max(best, m - 1)) { take it; niters = RANSACUpdateNumIters(...); }
// replayed in order by one lane, so that the order-dependent shrinking of niters
// comes out as in the sequential program.  One wavefront per problem; the rows'
// counts are brought into LDS by all lanes first.  Progress is parked in the
// result block between launches (R_ITERS, R_NINL, R_HYP, R_Q, R_NITERS, R_STATE).
constexpr int REPLAY_ROWS = 128;
__global__ void __launch_bounds__(64) ransac_replay_kernel(const RansacProb* __restrict__ probs, int row0, int nrows,
                                                           int max_iters, double confidence)
{
    __shared__ int32_t s_counts[REPLAY_ROWS * MAX_MODELS];
    const RansacProb pb = probs[blockIdx.x];
    const int lane = threadIdx.x;
    int32_t* r = pb.result;
    const int n = *pb.n_ptr, m = pb.m;
    int state = row0 == 0 ? 0 : r[R_STATE];
    if (row0 == 0 && n < m) state = R_STATE_DONE | R_STATE_FAILED;
    if (row0 > 0 && (state & R_STATE_DONE)) return;
    if (!(state & R_STATE_DONE))
        for (int i = lane; i < nrows * MAX_MODELS; i += 64) s_counts[i] = pb.counts[(size_t)row0 * MAX_MODELS + i];
    __syncthreads();
    if (lane != 0) return;
    int niters = row0 == 0 ? max_iters : r[R_NITERS];
    int iter = row0 == 0 ? 0 : r[R_ITERS];
    int best = row0 == 0 ? 0 : r[R_NINL];
    int hyp = row0 == 0 ? -1 : r[R_HYP];
    int q = row0 == 0 ? -1 : r[R_Q];
    if (!(state & R_STATE_DONE)) {
        const int valid = r[R_VALID];
        if (r[R_DIRECT]) {
            // n == m: one direct solve, every point an inlier
            state = R_STATE_DONE | (s_counts[0] < 0 ? R_STATE_FAILED : 0);
            hyp = 0;
            q = 0;
            best = n;
            iter = 0;
        } else {
            for (; iter < niters && iter < row0 + nrows; ++iter) {
                if (iter >= valid) {  // getSubset failed
                    if (iter == 0) state |= R_STATE_FAILED;
                    state |= R_STATE_DONE;
                    break;
                }
                const int32_t* row = s_counts + (iter - row0) * MAX_MODELS;
                for (int k = 0; k < MAX_MODELS; ++k) {
                    const int good = row[k];
                    if (good < 0) continue;
                    if (good > max(best, m - 1)) {
                        best = good;
                        hyp = iter;
                        q = k;
                        niters = update_num_iters_dev(confidence, (double)(n - good) / n, m, niters);
                    }
                }
            }
            if (iter >= niters) state |= R_STATE_DONE;
        }
    }
    if (state & R_STATE_DONE) {
        const bool ok = !(state & R_STATE_FAILED) && best > 0 && hyp >= 0;
        r[R_OK] = ok ? 1 : 0;
        r[R_HYP] = ok ? hyp : -1;
        r[R_Q] = ok ? q : -1;
        r[R_NINL] = ok ? best : 0;
    } else {
        r[R_HYP] = hyp;
        r[R_Q] = q;
        r[R_NINL] = best;
    }
    r[R_ITERS] = iter;
    r[R_NITERS] = niters;
    r[R_STATE] = state;
}

// Inlier mask of the selected model (result[R_HYP], result[R_Q]) and a copy of
// the model itself.  Leaves mask untouched when no model was found.
__global__ void __launch_bounds__(256) ransac_mask_kernel(const RansacProb* __restrict__ probs)
{
    const RansacProb pb = probs[blockIdx.y];
    const int n = *pb.n_ptr;
    if (!pb.result[R_OK]) return;
    const double* Mg = pb.models + (size_t)pb.result[R_HYP] * 90 + pb.result[R_Q] * 9;
    double M[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) M[i] = Mg[i];
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < 9) pb.model_out[i] = M[i];
    if (i >= n) return;
    int f;
    if (pb.result[R_DIRECT]) {
        f = 1;
    } else if (pb.model == SFM_GPU_MODEL_ESSENTIAL) {
        f = sampson_inlier(M, pb.x1[2 * i], pb.x1[2 * i + 1], pb.x2[2 * i], pb.x2[2 * i + 1], pb.tf,
                           pb.tlo, pb.thi);
    } else {
        f = symepi_inlier(M, pb.f1[2 * i], pb.f1[2 * i + 1], pb.f2[2 * i], pb.f2[2 * i + 1], pb.tf, pb.tlo, pb.thi);
    }
    pb.mask[i] = (uint8_t)f;
}

}  // namespace

hipError_t launch_ransac_prep(hipStream_t s, const RansacProb* probs, int n_prob, int cap)
{
    hipLaunchKernelGGL(ransac_prep_kernel, dim3((cap + 255) / 256 > 0 ? (cap + 255) / 256 : 1, n_prob),
                       dim3(256), 0, s, probs);
    return hipGetLastError();
}
hipError_t launch_ransac_table(hipStream_t s, const RansacProb* probs, int n_prob, int row0, int nrows)
{
    hipLaunchKernelGGL(ransac_table_kernel, dim3(n_prob), dim3(64), 0, s, probs, row0, nrows);
    return hipGetLastError();
}
hipError_t launch_ransac_table_fixed(hipStream_t s, const RansacProb* probs, int n_prob, int nrows)
{
    hipLaunchKernelGGL(ransac_table_fixed_kernel, dim3((nrows + 255) / 256, n_prob), dim3(256), 0, s, probs, nrows);
    return hipGetLastError();
}
hipError_t launch_ransac_solve(hipStream_t s, const RansacProb* probs, int n_prob, int model, int hyp0, int n_hyp,
                               int gen_rows, int split_mode)
{
    if (model == SFM_GPU_MODEL_FUNDAMENTAL)
    {
        // One lane per hypothesis needs 32 KB of LDS per wavefront, five wavefronts per
        // CU: a batch with more wavefronts than that at once runs in rounds, and beside
        // the descriptor conversion of the next batch (bench.py) it takes 100 us where
        // two lanes per hypothesis take 87 (alone 52 against 61: the split pays in
        // hand-overs).  Small batches stay with one lane (23 against 38 us per wavefront).
#ifndef SFM_SOLVE7_LPH
#define SFM_SOLVE7_LPH 2
#endif
        const bool split = SFM_SOLVE7_LPH > 1 &&
                           (split_mode < 0 ? (size_t)((n_hyp + 63) / 64) * (size_t)n_prob >= 1280 : split_mode > 0);
        if (!split) {
            hipLaunchKernelGGL(ransac_solve7_kernel, dim3((n_hyp + 63) / 64, n_prob), dim3(64), 0, s, probs, hyp0,
                               n_hyp, gen_rows);
        } else {
            constexpr int LPH = SFM_SOLVE7_LPH > 1 ? SFM_SOLVE7_LPH : 2, HPW = 64 / LPH;
            hipLaunchKernelGGL((ransac_solve7_split_kernel<LPH>), dim3((n_hyp + HPW - 1) / HPW, n_prob), dim3(64), 0,
                               s, probs, hyp0, n_hyp, gen_rows);
        }
        hipLaunchKernelGGL(ransac_roots7_parts_kernel, dim3((n_hyp + 63) / 64, n_prob), dim3(192), 0, s, probs, hyp0,
                           n_hyp);
    }
    else {
        hipLaunchKernelGGL(ransac_solve_kernel, dim3(n_hyp, n_prob), dim3(64), 0, s, probs, hyp0);
        hipLaunchKernelGGL(ransac_roots5_kernel, dim3((n_hyp + 3) / 4, n_prob), dim3(64), 0, s, probs, hyp0, n_hyp);
    }
    return hipGetLastError();
}
hipError_t launch_ransac_score(hipStream_t s, const RansacProb* probs, int n_prob, int cap, int model, int hyp0,
                               int n_hyp, bool staged, int filter_mode, int prune_mode)
{
    const int maxm = model == SFM_GPU_MODEL_ESSENTIAL ? 10 : 3;
    hipLaunchKernelGGL(ransac_compact_kernel, dim3(n_prob), dim3(1024), 0, s, probs, hyp0, n_hyp, maxm);
    const int chunk = 512;
    const int n_chunks = (cap + chunk - 1) / chunk > 0 ? (cap + chunk - 1) / chunk : 1;
    const unsigned groups = (unsigned)(((size_t)n_hyp * maxm + 63) / 64);
    // strided: the grid is smaller than the work and its workgroups loop
    auto score = [&](int chunk_pts, int slot_begin, int slot_end, int chunks, unsigned gx, int list_limit,
                     bool strided, int live_slot = -1) {
        const unsigned gw = (gx + SCORE_WAVES - 1) / SCORE_WAVES;
        // (a stage, live_slot >= 0: `chunks` work items per problem, shared out on the device)
        const dim3 grid = live_slot >= 0 ? dim3(n_prob, (chunks + SCORE_WAVES - 1) / SCORE_WAVES, 1)
                                         : (live_slot == -2 ? dim3(n_prob, chunks, gw) : dim3(gw, chunks, n_prob));
        const dim3 block(SCORE_WAVES * 64);
        const int f = filter_mode;
        if (model == SFM_GPU_MODEL_ESSENTIAL) {
            if (strided)
                hipLaunchKernelGGL((ransac_score_kernel<SFM_GPU_MODEL_ESSENTIAL, true>), grid, block, 0, s, probs,
                                   chunk_pts, f, slot_begin, slot_end, list_limit, live_slot);
            else
                hipLaunchKernelGGL((ransac_score_kernel<SFM_GPU_MODEL_ESSENTIAL, false>), grid, block, 0, s, probs,
                                   chunk_pts, f, slot_begin, slot_end, list_limit, live_slot);
        } else {
            if (strided)
                hipLaunchKernelGGL((ransac_score_kernel<SFM_GPU_MODEL_FUNDAMENTAL, true>), grid, block, 0, s, probs,
                                   chunk_pts, f, slot_begin, slot_end, list_limit, live_slot);
            else
                hipLaunchKernelGGL((ransac_score_kernel<SFM_GPU_MODEL_FUNDAMENTAL, false>), grid, block, 0, s, probs,
                                   chunk_pts, f, slot_begin, slot_end, list_limit, live_slot);
        }
    };
    // Stages with cuts.  Every model runs over an opening stretch; the list is
    // then put in order of the counts there and its best few are completed: their
    // largest count is the bound a model has to be able to reach.  Four stages
    // follow, their ends chosen per problem from that bound
    // (ransac_sortplan_kernel).  Nothing is compacted between them: in the ordered
    // list the models that have fallen out of reach sit together, so whole
    // wavefronts of them stop at the door, and a count of the leading groups
    // still in play keeps most of those from looking at their models at all.
    // (a batch of few problems has too few wavefronts in the later stages to hide
    // their start-up, and one workgroup ordering a long list is slow: such a batch,
    // e.g. one 32768-row pair with 4096 five-point hypotheses, takes the schedule below)
    if (staged && prune_mode == 1 && n_prob < 8) prune_mode = 2;
    if (prune_mode == 4) prune_mode = 1;  // tests: the ordered stages whatever the batch
    if (staged && prune_mode == 1 && cap >= 1024) {
        score(SCORE_OPENING / 2, -1, R_S0, 2, groups, INT_MAX, false, -2);  // (group-major: 75 against 68 us)
        hipLaunchKernelGGL(ransac_sortplan_kernel, dim3(n_prob), dim3(1024), 0, s, probs);
        // work items per problem and stage: one per group of the list at least (every group
        // may still be in play), and enough over the batch to fill the chip when few are
        const int items = std::max((int)groups, std::min(1024, SCORE_STAGE_ITEMS / n_prob));
        score(0, R_S0, R_S1, items, groups, INT_MAX, false, R_LIVE);
        score(0, R_S1, R_MID, items, groups, INT_MAX, false, R_LIVE + 1);
        score(0, R_MID, R_S3, items, groups, INT_MAX, false, R_LIVE + 2);
        score(0, R_S3, -1, items, groups, INT_MAX, false, R_LIVE + 3);
    } else if (staged && prune_mode >= 2 && cap >= 1024) {
        // Stages sized from the opening stretch's best count, the list compacted at a
        // cut between them (prune_mode 2: two cuts, 3: one).
        score(SCORE_OPENING, -1, R_S0, 1, groups, INT_MAX, false);
        // the first stage may take up to a quarter of the capacity: with the ratio
        // test leaving well under half of the queries, that covers outlier shares
        // up to about one half, beyond which the bound has nothing to drop anyway
        const int cover = std::max(2 * SCORE_OPENING, (cap / 4 + 255) / 256 * 256);
        hipLaunchKernelGGL(ransac_plan_kernel, dim3(n_prob), dim3(1024), 0, s, probs, cover, prune_mode == 3 ? 1 : 0);
        score(0, R_S0, R_S1, SCORE_SPLIT, groups, INT_MAX, false);
        hipLaunchKernelGGL(ransac_prune_kernel, dim3(n_prob), dim3(1024), 0, s, probs, (int)R_S1);
        // the survivors are few: pieces of 128 points, and a grid an eighth of
        // the list wide (a sixteenth after the second cut) that strides over
        // whatever is left
        const int depth = std::max(1, std::min(32, cap / 1024));
        if (prune_mode == 2) {
            score(128, R_S1, R_MID, depth, (groups + 7) / 8, INT_MAX, true);
            hipLaunchKernelGGL(ransac_prune_kernel, dim3(n_prob), dim3(1024), 0, s, probs, (int)R_MID);
            score(128, R_MID, -1, std::min(32, 2 * depth), (groups + 15) / 16, INT_MAX, true);
        } else {
            score(128, R_S1, -1, std::min(32, 2 * depth), (groups + 7) / 8, INT_MAX, true);
        }
    } else {
        score(chunk, -1, -1, n_chunks, groups, INT_MAX, false);
    }
    return hipGetLastError();
}
hipError_t launch_ransac_select(hipStream_t s, const RansacProb* probs, int n_prob, int n_hyp, const PairDev* pairs)
{
    hipLaunchKernelGGL(ransac_select_kernel, dim3(n_prob), dim3(1024), 0, s, probs, n_hyp, pairs);
    return hipGetLastError();
}
hipError_t launch_ransac_replay(hipStream_t s, const RansacProb* probs, int n_prob, int row0, int nrows, int max_iters,
                                double confidence)
{
    if (nrows > REPLAY_ROWS) return hipErrorInvalidValue;
    hipLaunchKernelGGL(ransac_replay_kernel, dim3(n_prob), dim3(64), 0, s, probs, row0, nrows, max_iters, confidence);
    return hipGetLastError();
}
hipError_t read_score_stamps(unsigned long long* out, int n_max, int* count)
{
#ifdef SFM_GPU_TIMING_BUILD
    unsigned per[64];
    hipError_t e = hipMemcpyFromSymbol(per, HIP_SYMBOL(g_score_stamp_count), sizeof(per));
    if (e != hipSuccess) return e;
    // slots are 64 * k + b for k < per[b]: copy the covered prefix and keep the slots that were written
    unsigned hi = 0;
    for (int b = 0; b < 64; ++b) hi = std::max(hi, per[b]);
    const unsigned span = std::min(64u * hi, SCORE_STAMPS);
    std::vector<unsigned long long> all((size_t)span * 4);
    if (span) {
        e = hipMemcpyFromSymbol(all.data(), HIP_SYMBOL(g_score_stamps), all.size() * 8);
        if (e != hipSuccess) return e;
    }
    int c = 0;
    for (unsigned slot = 0; slot < span && c < n_max; ++slot)
        if (slot / 64u < per[slot % 64u]) {
            for (int w = 0; w < 4; ++w) out[4 * (size_t)c + w] = all[4 * (size_t)slot + w];
            ++c;
        }
    *count = c;
    unsigned zero[64] = {};
    return hipMemcpyToSymbol(HIP_SYMBOL(g_score_stamp_count), zero, sizeof(zero));
#else
    (void)out; (void)n_max; (void)count;
    return hipErrorNotSupported;
#endif
}

hipError_t launch_ransac_mask(hipStream_t s, const RansacProb* probs, int n_prob, int cap)
{
    hipLaunchKernelGGL(ransac_mask_kernel, dim3((cap + 255) / 256 > 0 ? (cap + 255) / 256 : 1, n_prob),
                       dim3(256), 0, s, probs);
    return hipGetLastError();
}

}  // namespace sfm
