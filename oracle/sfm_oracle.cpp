/*
 * CPU ORACLE — TEST INFRASTRUCTURE ONLY.  PARITY UNPINNED (see sfm_oracle.h).
 *
 * Plain C++17, no dependencies.  Build with -ffp-contract=off: several results
 * below are compared bit-for-bit with the GPU path, and that requires that no
 * multiply-add is fused on either side.
 *
 * What each part restates:
 *   knn            cv::BFMatcher(NORM_L2)::knnMatch as called at
 *                  /root/reference/main.cpp:184 (arithmetic: OpenCV 4.5.5
 *                  core batchDistance: sqrtf of a float32 sum of squared
 *                  differences, then a K-slot insertion scan in ascending
 *                  train order with strict comparisons)
 *   ratio test     /root/reference/main.cpp:189-197
 *   rng/getSubset  cv::RNG and RANSACPointSetRegistrator::getSubset (OpenCV
 *                  4.5.5 calib3d ptsetreg.cpp), reached from main.cpp:224
 *   ransac         RANSACPointSetRegistrator::run + RANSACUpdateNumIters
 *   five_point     Nister, "An efficient solution to the five-point relative
 *                  pose problem", PAMI 2004 (the method behind
 *                  cv::findEssentialMat, main.cpp:224)
 *   sampson        EMEstimatorCallback::computeError (five-point.cpp)
 *   seven_point    FMEstimatorCallback (fundam.cpp): 7-point solver and the
 *                  symmetric epipolar distance of findFundamentalMat
 *
 * Deliberate, documented deviations where OpenCV's result is implementation
 * defined or cannot be known without its source:
 *   - null spaces come from Gauss-Jordan elimination with complete pivoting,
 *     not from cv::SVD: the subspace is the same, the basis inside it is
 *     arbitrary in both cases;
 *   - polynomial real roots come from bracketing by derivative roots plus a
 *     safeguarded Newton iteration, not cv::solvePoly; candidate models are
 *     emitted in ascending root order (OpenCV's order is solvePoly's).
 */
#include "sfm_oracle.h"

#include <algorithm>
#include <cfloat>
#include <climits>
#include <cstdint>
#include <cmath>
#include <cstring>
#include <thread>
#include <vector>

namespace {

// ---------------------------------------------------------------- k-NN ----

void knn_rows(const float* query, int q0, int q1, size_t qstep,
              const float* train, int nt, size_t tstep, int dim, int k,
              oracle_dmatch* out, int32_t* counts)
{
    const int K = std::min(k, nt);
    std::vector<float> buf(nt > 0 ? nt : 1);
    std::vector<float> dist(k > 0 ? k : 1);
    std::vector<int> nidx(k > 0 ? k : 1);
    for (int i = q0; i < q1; i++) {
        const float* q = (const float*)((const char*)query + (size_t)i * qstep);
        for (int j = 0; j < nt; j++) {
            const float* t = (const float*)((const char*)train + (size_t)j * tstep);
            float s = 0.f;
            for (int d = 0; d < dim; d++) {
                float df = q[d] - t[d];
                s += df * df;
            }
            buf[j] = std::sqrt(s);
        }
        for (int s = 0; s < k; s++) { dist[s] = FLT_MAX; nidx[s] = -1; }
        if (K > 0) {
            for (int j = 0; j < nt; j++) {
                float d = buf[j];
                if (d < dist[K - 1]) {
                    int p = K - 2;
                    for (; p >= 0 && dist[p] > d; p--) {
                        dist[p + 1] = dist[p];
                        nidx[p + 1] = nidx[p];
                    }
                    dist[p + 1] = d;
                    nidx[p + 1] = j;
                }
            }
        }
        int cnt = 0;
        for (int s = 0; s < k; s++) {
            oracle_dmatch& m = out[(size_t)i * k + s];
            m.queryIdx = i;
            m.imgIdx = 0;
            if (s < K && nidx[s] >= 0) {
                m.trainIdx = nidx[s];
                m.distance = dist[s];
                cnt++;
            } else {
                m.trainIdx = -1;
                m.distance = FLT_MAX;
            }
        }
        counts[i] = cnt;
    }
}

// The same rows for the timing leg of the benchmark: descriptors that hold
// integers 0..255 (what cv::SIFT emits) are narrowed to bytes once and the sum
// of squared differences is taken in int32, four query rows against one train
// row at a time.  The float32 sum knn_rows() forms is exact for such inputs
// (< 2^24) whatever the order of the additions, so s below converts to the very
// float knn_rows() passes to sqrt, and the insertion scan is the same scan: a
// candidate with s >= the integer behind the current K-th distance has
// sqrtf(s) >= that distance and is rejected there too.  Returns false (nothing
// written) when an input value is not an integer in 0..255.
bool knn_rows_u8(const uint8_t* qb, int q0, int q1, const uint8_t* tb, int nt, int dim, int k, oracle_dmatch* out,
                 int32_t* counts)
{
    const int K = std::min(k, nt);
    constexpr int QB = 4;
    std::vector<float> dist((size_t)QB * (k > 0 ? k : 1));
    std::vector<int> nidx((size_t)QB * (k > 0 ? k : 1));
    std::vector<int32_t> lim(QB);
    for (int i0 = q0; i0 < q1; i0 += QB) {
        const int nb = std::min(QB, q1 - i0);
        const uint8_t* q[QB];
        for (int b = 0; b < QB; b++) q[b] = qb + (size_t)(i0 + std::min(b, nb - 1)) * dim;
        for (int b = 0; b < QB; b++) {
            for (int s = 0; s < k; s++) { dist[(size_t)b * k + s] = FLT_MAX; nidx[(size_t)b * k + s] = -1; }
            lim[b] = INT32_MAX;   // integer behind dist[K-1]: INT32_MAX while that slot is empty
        }
        for (int j = 0; j < nt && K > 0; j++) {
            const uint8_t* t = tb + (size_t)j * dim;
            int32_t s0 = 0, s1 = 0, s2 = 0, s3 = 0;
            for (int d = 0; d < dim; d++) {
                const int tv = t[d];
                const int d0 = q[0][d] - tv, d1 = q[1][d] - tv, d2 = q[2][d] - tv, d3 = q[3][d] - tv;
                s0 += d0 * d0;
                s1 += d1 * d1;
                s2 += d2 * d2;
                s3 += d3 * d3;
            }
            const int32_t ss[QB] = {s0, s1, s2, s3};
            for (int b = 0; b < nb; b++) {
                if (ss[b] >= lim[b]) continue;
                const float dv = std::sqrt((float)ss[b]);
                float* db = &dist[(size_t)b * k];
                int* nb_ = &nidx[(size_t)b * k];
                if (dv < db[K - 1]) {
                    int p = K - 2;
                    for (; p >= 0 && db[p] > dv; p--) {
                        db[p + 1] = db[p];
                        nb_[p + 1] = nb_[p];
                    }
                    db[p + 1] = dv;
                    nb_[p + 1] = j;
                    // smallest integer whose float root reaches the K-th distance: every
                    // later candidate at or above it is rejected by the float compare too
                    if (nb_[K - 1] >= 0) {
                        const float kth = db[K - 1];
                        int32_t L = (int32_t)((double)kth * (double)kth);
                        while (L > 0 && std::sqrt((float)(L - 1)) >= kth) --L;
                        while (std::sqrt((float)L) < kth) ++L;
                        lim[b] = L;
                    }
                }
            }
        }
        for (int b = 0; b < nb; b++) {
            const int i = i0 + b;
            int cnt = 0;
            for (int s = 0; s < k; s++) {
                oracle_dmatch& m = out[(size_t)i * k + s];
                m.queryIdx = i;
                m.imgIdx = 0;
                if (s < K && nidx[(size_t)b * k + s] >= 0) {
                    m.trainIdx = nidx[(size_t)b * k + s];
                    m.distance = dist[(size_t)b * k + s];
                    cnt++;
                } else {
                    m.trainIdx = -1;
                    m.distance = FLT_MAX;
                }
            }
            counts[i] = cnt;
        }
    }
    return true;
}

// rows of float32 -> bytes; false when a value is not an integer in 0..255
bool narrow_rows(const float* src, int n, size_t step, int dim, std::vector<uint8_t>& dst)
{
    dst.resize((size_t)std::max(n, 1) * dim);
    for (int i = 0; i < n; i++) {
        const float* r = (const float*)((const char*)src + (size_t)i * step);
        for (int d = 0; d < dim; d++) {
            const float v = r[d];
            if (!(v >= 0.f && v <= 255.f) || v != (float)(int)v) return false;
            dst[(size_t)i * dim + d] = (uint8_t)(int)v;
        }
    }
    return true;
}

// ------------------------------------------------------------- cv::RNG ----

struct CvRng {
    uint64_t state;
    explicit CvRng(uint64_t s) : state(s ? s : 0xffffffffULL) {}
    uint32_t next()
    {
        state = (uint64_t)(uint32_t)state * 4164903690U + (uint32_t)(state >> 32);
        return (uint32_t)state;
    }
    int uniform(int a, int b) { return a == b ? a : (int)(next() % (uint32_t)(b - a) + a); }
};

// ----------------------------------------------- small dense linear algebra

// Null space of a rows x 9 system by Gauss-Jordan with complete pivoting.
// On success basis holds (9-rows) vectors of 9 entries.  Pivot search scans
// row-major and keeps the first maximum.
bool null_space_9(const double* Ain, int rows, double* basis)
{
    double A[7][9];
    int perm[9];
    for (int i = 0; i < rows; i++)
        for (int j = 0; j < 9; j++) A[i][j] = Ain[i * 9 + j];
    for (int j = 0; j < 9; j++) perm[j] = j;
    for (int p = 0; p < rows; p++) {
        int bi = p, bj = p;
        double best = -1.0;
        for (int i = p; i < rows; i++)
            for (int j = p; j < 9; j++) {
                double v = std::fabs(A[i][j]);
                if (v > best) { best = v; bi = i; bj = j; }
            }
        if (!(best > 0.0)) return false;
        if (bi != p)
            for (int j = 0; j < 9; j++) std::swap(A[p][j], A[bi][j]);
        if (bj != p) {
            for (int i = 0; i < rows; i++) std::swap(A[i][p], A[i][bj]);
            std::swap(perm[p], perm[bj]);
        }
        double piv = A[p][p];
        for (int j = p + 1; j < 9; j++) A[p][j] = A[p][j] / piv;
        A[p][p] = 1.0;
        for (int i = 0; i < rows; i++) {
            if (i == p) continue;
            double f = A[i][p];
            for (int j = p + 1; j < 9; j++) A[i][j] = A[i][j] - f * A[p][j];
            A[i][p] = 0.0;
        }
    }
    const int nfree = 9 - rows;
    for (int k = 0; k < nfree; k++) {
        double* v = basis + k * 9;
        for (int j = 0; j < 9; j++) v[j] = 0.0;
        v[perm[rows + k]] = 1.0;
        for (int i = 0; i < rows; i++) v[perm[i]] = -A[i][rows + k];
    }
    return true;
}

double horner(const double* c, int n, double x)
{
    double s = c[n];
    for (int i = n - 1; i >= 0; i--) s = s * x + c[i];
    return s;
}

// Root of polynomial P (degree n, derivative dP of degree n-1) inside the
// bracket (lo,hi) across which P changes sign; lo_neg says P(lo) < 0.
// Newton steps, replaced by a bisection of the bracket whenever the Newton
// step would leave the bracket or shrinks slower than by half.  Runs to a
// fixed point (no tolerance), so the result depends on IEEE arithmetic only.
double solve_bracket(const double* P, const double* dP, int n, double lo, double hi, bool lo_neg)
{
    double xl = lo_neg ? lo : hi;  // P(xl) < 0 < P(xh)
    double xh = lo_neg ? hi : lo;
    double x = 0.5 * (lo + hi);
    double dxold = std::fabs(hi - lo);
    double dx = dxold;
    double f = horner(P, n, x);
    double df = horner(dP, n - 1, x);
    for (int it = 0; it < 300; it++) {
        if (f == 0.0) return x;
        if (f < 0.0) xl = x; else xh = x;
        const bool bisect = (((x - xh) * df - f) * ((x - xl) * df - f) > 0.0) ||
                            (std::fabs(2.0 * f) > std::fabs(dxold * df));
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
        f = horner(P, n, x);
        df = horner(dP, n - 1, x);
    }
    return x;
}

// The first and the last interval of a level reach out to the root bound -B / B,
// which is usually far beyond the root they hold, and a Newton/bisection
// iteration started from the middle of such an interval spends dozens of steps
// coming back.  P is monotone there (no critical point beyond the outermost
// one), so the root is bracketed first by stepping away from the finite end in
// doubling steps (first step: 1/16 of max(1, |end|)).  Only done when P changes
// sign over the interval.
void tighten_open_interval(const double* P, int k, bool open_lo, bool open_hi, double& lo, double& hi,
                           double& flo, double& fhi)
{
    if (flo == 0.0 || fhi == 0.0 || ((flo < 0.0) == (fhi < 0.0))) return;
    if (open_hi) {
        double w = 0.0625 * std::max(1.0, std::fabs(lo));
        const double e = lo;
        double a = lo, fa = flo;
        for (;;) {
            const double x = e + w;
            if (!(x < hi)) break;
            const double fx = horner(P, k, x);
            if (fx == 0.0 || ((fx < 0.0) != (flo < 0.0))) { hi = x; fhi = fx; break; }
            a = x;
            fa = fx;
            w = w * 2.0;
        }
        lo = a;
        flo = fa;
    } else if (open_lo) {
        double w = 0.0625 * std::max(1.0, std::fabs(hi));
        const double e = hi;
        double b = hi, fb = fhi;
        for (;;) {
            const double x = e - w;
            if (!(x > lo)) break;
            const double fx = horner(P, k, x);
            if (fx == 0.0 || ((fx < 0.0) != (fhi < 0.0))) { lo = x; flo = fx; break; }
            b = x;
            fb = fx;
            w = w * 2.0;
        }
        hi = b;
        fhi = fb;
    }
}

// Real roots, ascending, of q[0] + q[1] x + q[2] x^2 with q[2] != 0, by the
// cancellation-free form of the quadratic formula (unfused arithmetic,
// correctly rounded square root): none for a negative discriminant, the double
// root once for a zero discriminant.
int quadratic_roots(const double* q, double* r)
{
    const double a = q[2], b = q[1], c = q[0];
    const double disc = b * b - 4.0 * a * c;
    if (!(disc >= 0.0)) return 0;
    if (disc == 0.0) {
        r[0] = -b / (2.0 * a);
        return 1;
    }
    const double s = std::sqrt(disc);
    const double t = -0.5 * (b + (b < 0.0 ? -s : s));
    const double r1 = t / a, r2 = c / t;
    r[0] = r1 < r2 ? r1 : r2;
    r[1] = r1 < r2 ? r2 : r1;
    return 2;
}

// All real roots of c[0] + c[1] z + ... + c[n] z^n (n <= 10), ascending.
// Level k of the recursion isolates the roots of the (n-k)-th derivative
// between consecutive roots of the next derivative, where it is monotone.
int real_roots(const double* cin, int n, double* roots)
{
    double c[11];
    double cmax = 0.0;
    const bool small_degree = n <= 3;  // as given, before vanishing leading coefficients are dropped
    for (int i = 0; i <= n; i++) cmax = std::max(cmax, std::fabs(cin[i]));
    if (!(cmax > 0.0) || !std::isfinite(cmax)) return 0;
    for (int i = 0; i <= n; i++) c[i] = cin[i] / cmax;
    while (n > 0 && c[n] == 0.0) n--;
    if (n == 0) return 0;
    double D[11][11];
    for (int i = 0; i <= n; i++) D[n][i] = c[i];
    for (int k = n; k > 1; k--)
        for (int i = 0; i < k; i++) D[k - 1][i] = (double)(i + 1) * D[k][i + 1];
    double B = 0.0;
    for (int i = 0; i < n; i++) B = std::max(B, std::fabs(c[i] / c[n]));
    B = 1.0 + B;
    if (!(B < 1e25)) B = 1e25;
    double prev[10], cur[10];
    int mprev = 1, mcur = 0;
    prev[0] = -D[1][0] / D[1][1];
    if (n == 1) { roots[0] = prev[0]; return 1; }
    int k0 = 2;
    if (small_degree) {
        // Quadratics and cubics (the seven-point solver's polynomial): the
        // quadratic level has a closed form; the cubic level then brackets
        // between those critical points as every higher level does.
        mprev = quadratic_roots(D[2], prev);
        k0 = 3;
    }
    for (int k = k0; k <= n; k++) {
        // interval t = [e_t, e_{t+1}] with e_0 = -B, e_1..e_m the critical
        // points (clamped to [-B, B]), e_{m+1} = B; every interval is handled
        // independently of the others
        mcur = 0;
        for (int t = 0; t <= mprev; t++) {
            double lo = (t == 0) ? -B : prev[t - 1];
            double hi = (t < mprev) ? prev[t] : B;
            if (lo < -B) lo = -B;
            if (lo > B) lo = B;
            if (hi < -B) hi = -B;
            if (hi > B) hi = B;
            if (!(hi > lo)) continue;
            double flo = horner(D[k], k, lo);
            double fhi = horner(D[k], k, hi);
            tighten_open_interval(D[k], k, t == 0 && mprev >= 1, t == mprev && mprev >= 1, lo, hi, flo, fhi);
            if (flo == 0.0) {
                if (mcur < 10) cur[mcur++] = lo;
            } else if (fhi != 0.0 && ((flo < 0.0) != (fhi < 0.0))) {
                if (mcur < 10) cur[mcur++] = solve_bracket(D[k], D[k - 1], k, lo, hi, flo < 0.0);
            } else if (fhi == 0.0 && t == mprev) {
                if (mcur < 10) cur[mcur++] = hi;
            }
        }
        for (int i = 0; i < mcur; i++) prev[i] = cur[i];
        mprev = mcur;
    }
    for (int i = 0; i < mprev; i++) roots[i] = prev[i];
    return mprev;
}

// ---------------------------------------------------- five-point (Nister) --

// Monomial bookkeeping for polynomials in (x,y,z).
//   degree <= 1: [x, y, z, 1]
//   degree <= 2: [x2, xy, xz, y2, yz, z2, x, y, z, 1]
//   degree <= 3: Nister's elimination order
//     [x3, y3, x2y, xy2, x2z, x2, y2z, y2, xyz, xy,
//      xz2, xz, x, yz2, yz, y, z3, z2, z, 1]
const int E1[4][3] = {{1,0,0},{0,1,0},{0,0,1},{0,0,0}};
const int E2[10][3] = {{2,0,0},{1,1,0},{1,0,1},{0,2,0},{0,1,1},{0,0,2},{1,0,0},{0,1,0},{0,0,1},{0,0,0}};
const int E3[20][3] = {{3,0,0},{0,3,0},{2,1,0},{1,2,0},{2,0,1},{2,0,0},{0,2,1},{0,2,0},{1,1,1},{1,1,0},
                       {1,0,2},{1,0,1},{1,0,0},{0,1,2},{0,1,1},{0,1,0},{0,0,3},{0,0,2},{0,0,1},{0,0,0}};

struct MonoTables {
    int t11[4][4];
    int t21[10][4];
    MonoTables()
    {
        for (int i = 0; i < 4; i++)
            for (int j = 0; j < 4; j++) {
                int e[3] = {E1[i][0] + E1[j][0], E1[i][1] + E1[j][1], E1[i][2] + E1[j][2]};
                t11[i][j] = -1;
                for (int m = 0; m < 10; m++)
                    if (E2[m][0] == e[0] && E2[m][1] == e[1] && E2[m][2] == e[2]) t11[i][j] = m;
            }
        for (int i = 0; i < 10; i++)
            for (int j = 0; j < 4; j++) {
                int e[3] = {E2[i][0] + E1[j][0], E2[i][1] + E1[j][1], E2[i][2] + E1[j][2]};
                t21[i][j] = -1;
                for (int m = 0; m < 20; m++)
                    if (E3[m][0] == e[0] && E3[m][1] == e[1] && E3[m][2] == e[2]) t21[i][j] = m;
            }
    }
};
const MonoTables MT;

void mul11(const double* a, const double* b, double* out)
{
    for (int m = 0; m < 10; m++) out[m] = 0.0;
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++) out[MT.t11[i][j]] = out[MT.t11[i][j]] + a[i] * b[j];
}

void mul21(const double* a, const double* b, double* out)
{
    for (int m = 0; m < 20; m++) out[m] = 0.0;
    for (int i = 0; i < 10; i++)
        for (int j = 0; j < 4; j++) out[MT.t21[i][j]] = out[MT.t21[i][j]] + a[i] * b[j];
}

// out[0..na+nb] = a * b, polynomials in z with c[i] the coefficient of z^i
void pmul(const double* a, int na, const double* b, int nb, double* out)
{
    for (int i = 0; i <= na + nb; i++) out[i] = 0.0;
    for (int i = 0; i <= na; i++)
        for (int j = 0; j <= nb; j++) out[i + j] = out[i + j] + a[i] * b[j];
}

int five_point(const double* x1, const double* x2, double* models)
{
    // epipolar constraint rows for x2^T E x1 = 0, E row-major
    double Q[5 * 9];
    for (int i = 0; i < 5; i++) {
        double u1 = x1[2 * i], v1 = x1[2 * i + 1], u2 = x2[2 * i], v2 = x2[2 * i + 1];
        double* q = Q + 9 * i;
        q[0] = u2 * u1; q[1] = u2 * v1; q[2] = u2;
        q[3] = v2 * u1; q[4] = v2 * v1; q[5] = v2;
        q[6] = u1;      q[7] = v1;      q[8] = 1.0;
    }
    double N[4 * 9];  // X, Y, Z, W
    if (!null_space_9(Q, 5, N)) return 0;

    // e[r][c] = polynomial (x,y,z,1) of entry (r,c) of E = xX + yY + zZ + W
    double e[3][3][4];
    for (int r = 0; r < 3; r++)
        for (int c = 0; c < 3; c++)
            for (int k = 0; k < 4; k++) e[r][c][k] = N[k * 9 + r * 3 + c];

    double A[10][20];
    // EE^T (symmetric), trace, Lambda = EE^T - 0.5 tr I
    double L[3][3][10];
    for (int i = 0; i < 3; i++)
        for (int j = i; j < 3; j++) {
            double t0[10], t1[10], t2[10];
            mul11(e[i][0], e[j][0], t0);
            mul11(e[i][1], e[j][1], t1);
            mul11(e[i][2], e[j][2], t2);
            for (int m = 0; m < 10; m++) L[i][j][m] = (t0[m] + t1[m]) + t2[m];
        }
    for (int m = 0; m < 10; m++) {
        double tr = (L[0][0][m] + L[1][1][m]) + L[2][2][m];
        double h = 0.5 * tr;
        L[0][0][m] = L[0][0][m] - h;
        L[1][1][m] = L[1][1][m] - h;
        L[2][2][m] = L[2][2][m] - h;
    }
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < i; j++)
            for (int m = 0; m < 10; m++) L[i][j][m] = L[j][i][m];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) {
            double t0[20], t1[20], t2[20];
            mul21(L[i][0], e[0][j], t0);
            mul21(L[i][1], e[1][j], t1);
            mul21(L[i][2], e[2][j], t2);
            for (int m = 0; m < 20; m++) A[i * 3 + j][m] = (t0[m] + t1[m]) + t2[m];
        }
    {
        double a[10], b[10], d0[10], d1[10], d2[10];
        mul11(e[1][1], e[2][2], a); mul11(e[1][2], e[2][1], b);
        for (int m = 0; m < 10; m++) d0[m] = a[m] - b[m];
        mul11(e[1][0], e[2][2], a); mul11(e[1][2], e[2][0], b);
        for (int m = 0; m < 10; m++) d1[m] = a[m] - b[m];
        mul11(e[1][0], e[2][1], a); mul11(e[1][1], e[2][0], b);
        for (int m = 0; m < 10; m++) d2[m] = a[m] - b[m];
        double t0[20], t1[20], t2[20];
        mul21(d0, e[0][0], t0);
        mul21(d1, e[0][1], t1);
        mul21(d2, e[0][2], t2);
        for (int m = 0; m < 20; m++) A[9][m] = (t0[m] - t1[m]) + t2[m];
    }

    // Gauss-Jordan on the 10 leading columns, partial pivoting (first max)
    for (int p = 0; p < 10; p++) {
        int bi = p;
        double best = -1.0;
        for (int i = p; i < 10; i++) {
            double v = std::fabs(A[i][p]);
            if (v > best) { best = v; bi = i; }
        }
        if (!(best > 0.0)) return 0;
        if (bi != p)
            for (int j = 0; j < 20; j++) std::swap(A[p][j], A[bi][j]);
        double piv = A[p][p];
        for (int j = p + 1; j < 20; j++) A[p][j] = A[p][j] / piv;
        A[p][p] = 1.0;
        for (int i = 0; i < 10; i++) {
            if (i == p) continue;
            double f = A[i][p];
            for (int j = p + 1; j < 20; j++) A[i][j] = A[i][j] - f * A[p][j];
            A[i][p] = 0.0;
        }
    }

    // rows e..j = 4..9; B(z) rows: e - z f, g - z h, i - z j
    // B[r][0], B[r][1]: degree 3 in z (coefficient of x, of y); B[r][2]: degree 4
    double B[3][3][5];
    for (int r = 0; r < 3; r++) {
        const double* hi = A[4 + 2 * r];  // has the extra factor z in its leading monomial
        const double* lo = A[5 + 2 * r];
        for (int v = 0; v < 2; v++) {
            int c0 = 10 + 3 * v;  // columns [v z^2, v z, v]
            B[r][v][0] = hi[c0 + 2];
            B[r][v][1] = hi[c0 + 1] - lo[c0 + 2];
            B[r][v][2] = hi[c0 + 0] - lo[c0 + 1];
            B[r][v][3] = -lo[c0 + 0];
            B[r][v][4] = 0.0;
        }
        // columns 16..19 = [z^3, z^2, z, 1]
        B[r][2][0] = hi[19];
        B[r][2][1] = hi[18] - lo[19];
        B[r][2][2] = hi[17] - lo[18];
        B[r][2][3] = hi[16] - lo[17];
        B[r][2][4] = -lo[16];
    }
    // det B(z), degree 10
    double c[11];
    {
        double p1[8], p2[8], m0[8], m1[8], m2[7], t0[11], t1[11], t2[11];
        pmul(B[1][1], 3, B[2][2], 4, p1); pmul(B[1][2], 4, B[2][1], 3, p2);
        for (int i = 0; i < 8; i++) m0[i] = p1[i] - p2[i];
        pmul(B[1][0], 3, B[2][2], 4, p1); pmul(B[1][2], 4, B[2][0], 3, p2);
        for (int i = 0; i < 8; i++) m1[i] = p1[i] - p2[i];
        pmul(B[1][0], 3, B[2][1], 3, p1); pmul(B[1][1], 3, B[2][0], 3, p2);
        for (int i = 0; i < 7; i++) m2[i] = p1[i] - p2[i];
        pmul(B[0][0], 3, m0, 7, t0);
        pmul(B[0][1], 3, m1, 7, t1);
        pmul(B[0][2], 4, m2, 6, t2);
        for (int i = 0; i < 11; i++) c[i] = (t0[i] - t1[i]) + t2[i];
    }
    double roots[10];
    int nr = real_roots(c, 10, roots);
    int count = 0;
    for (int k = 0; k < nr; k++) {
        double z = roots[k];
        double b[3][3];
        for (int r = 0; r < 3; r++) {
            b[r][0] = horner(B[r][0], 3, z);
            b[r][1] = horner(B[r][1], 3, z);
            b[r][2] = horner(B[r][2], 4, z);
        }
        // null vector of b by the row-pair cross product with the largest |n_z|
        double bestn[3] = {0, 0, 0};
        double bestz = -1.0;
        const int pr[3][2] = {{0, 1}, {0, 2}, {1, 2}};
        for (int q = 0; q < 3; q++) {
            const double* r0 = b[pr[q][0]];
            const double* r1 = b[pr[q][1]];
            double nx = r0[1] * r1[2] - r0[2] * r1[1];
            double ny = r0[2] * r1[0] - r0[0] * r1[2];
            double nz = r0[0] * r1[1] - r0[1] * r1[0];
            double az = std::fabs(nz);
            if (az > bestz) { bestz = az; bestn[0] = nx; bestn[1] = ny; bestn[2] = nz; }
        }
        double nn = std::sqrt((bestn[0] * bestn[0] + bestn[1] * bestn[1]) + bestn[2] * bestn[2]);
        if (!(nn > 0.0) || !(bestz / nn >= 1e-10)) continue;
        double x = bestn[0] / bestn[2];
        double y = bestn[1] / bestn[2];
        double E[9];
        double s = 0.0;
        for (int i = 0; i < 9; i++) {
            E[i] = ((x * N[i] + y * N[9 + i]) + z * N[18 + i]) + N[27 + i];
            s = s + E[i] * E[i];
        }
        double nrm = std::sqrt(s);
        if (!(nrm > 0.0) || !std::isfinite(nrm)) continue;
        for (int i = 0; i < 9; i++) models[count * 9 + i] = E[i] / nrm;
        count++;
    }
    return count;
}

// -------------------------------------------------------- seven-point F ----

int seven_point(const float* m1, const float* m2, double* models)
{
    double Q[7 * 9];
    for (int i = 0; i < 7; i++) {
        double x0 = m1[2 * i], y0 = m1[2 * i + 1], x1 = m2[2 * i], y1 = m2[2 * i + 1];
        double* q = Q + 9 * i;
        q[0] = x1 * x0; q[1] = x1 * y0; q[2] = x1;
        q[3] = y1 * x0; q[4] = y1 * y0; q[5] = y1;
        q[6] = x0;      q[7] = y0;      q[8] = 1.0;
    }
    double N[2 * 9];
    if (!null_space_9(Q, 7, N)) return 0;
    double f1[9], f2[9];
    for (int i = 0; i < 9; i++) { f2[i] = N[9 + i]; f1[i] = N[i] - f2[i]; }
    // det(lambda f1 + f2) = c3 l^3 + c2 l^2 + c1 l + c0 with F = l*(N0-N1) + N1
    // entries are degree-1 polynomials in lambda: p[i] = {f2[i], f1[i]}
    double p[9][2];
    for (int i = 0; i < 9; i++) { p[i][0] = f2[i]; p[i][1] = f1[i]; }
    double a[3], b[3], m0[3], mm1[3], mm2[3], t0[4], t1[4], t2[4], c[4];
    pmul(p[4], 1, p[8], 1, a); pmul(p[5], 1, p[7], 1, b);
    for (int i = 0; i < 3; i++) m0[i] = a[i] - b[i];
    pmul(p[3], 1, p[8], 1, a); pmul(p[5], 1, p[6], 1, b);
    for (int i = 0; i < 3; i++) mm1[i] = a[i] - b[i];
    pmul(p[3], 1, p[7], 1, a); pmul(p[4], 1, p[6], 1, b);
    for (int i = 0; i < 3; i++) mm2[i] = a[i] - b[i];
    pmul(p[0], 1, m0, 2, t0);
    pmul(p[1], 1, mm1, 2, t1);
    pmul(p[2], 1, mm2, 2, t2);
    for (int i = 0; i < 4; i++) c[i] = (t0[i] - t1[i]) + t2[i];
    double roots[3];
    int nr = real_roots(c, 3, roots);
    int count = 0;
    for (int k = 0; k < nr; k++) {
        double lambda = roots[k], mu = 1.0;
        double s = f1[8] * lambda + f2[8];
        double* F = models + 9 * count;
        if (std::fabs(s) > DBL_EPSILON) {
            mu = 1.0 / s;
            lambda = lambda * mu;
            F[8] = 1.0;
        } else {
            F[8] = 0.0;
        }
        for (int i = 0; i < 8; i++) F[i] = f1[i] * lambda + f2[i] * mu;
        count++;
    }
    return count;
}

bool have_collinear(const float* pts, const int* idx, int count)
{
    int i = count - 1;
    const float* pi = pts + 2 * idx[i];
    for (int j = 0; j < i; j++) {
        const float* pj = pts + 2 * idx[j];
        double dx1 = pj[0] - pi[0];
        double dy1 = pj[1] - pi[1];
        for (int k = 0; k < j; k++) {
            const float* pk = pts + 2 * idx[k];
            double dx2 = pk[0] - pi[0];
            double dy2 = pk[1] - pi[1];
            if (std::fabs(dx2 * dy1 - dy2 * dx1) <=
                FLT_EPSILON * (std::fabs(dx1) + std::fabs(dy1) + std::fabs(dx2) + std::fabs(dy2)))
                return true;
        }
    }
    return false;
}

// getSubset: m distinct uniform draws; optional subset check redraws the
// whole subset (at most max_attempts times).
bool get_subset(CvRng& rng, int count, int m, int* idx, const float* p1, const float* p2,
                int max_attempts)
{
    int iters = 0;
    int i = 0;
    for (; iters < max_attempts; iters++) {
        for (i = 0; i < m && iters < max_attempts;) {
            int v;
            for (;;) {
                v = idx[i] = rng.uniform(0, count);
                int j = 0;
                for (; j < i; j++)
                    if (v == idx[j]) break;
                if (j == i) break;
            }
            i++;
        }
        if (i == m && p1 && (have_collinear(p1, idx, m) || have_collinear(p2, idx, m))) continue;
        break;
    }
    return i == m && iters < max_attempts;
}

// Fixed-count mode sample table: every row is drawn independently of the others
// from a counter-based generator (splitmix64 finaliser over row / slot / attempt
// counters), so rows can be produced in any order or in parallel.  This mode is
// a benchmark mode with no OpenCV counterpart; the adaptive mode keeps OpenCV's
// sequential cv::RNG stream.
const uint64_t FIXED_TABLE_SEED = 0x53464D5F5441424CULL;

uint64_t mix64(uint64_t z)
{
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}

// index in [0,count) for (row, subset attempt u, slot i, slot attempt t)
int fixed_draw(int count, uint32_t row, uint32_t u, uint32_t i, uint32_t t)
{
    uint64_t ctr = ((uint64_t)row << 32) | ((uint64_t)(u & 0xFF) << 24) | ((uint64_t)(i & 0xFF) << 16) |
                   (uint64_t)(t & 0xFFFF);
    uint64_t h = mix64(FIXED_TABLE_SEED + ctr * 0x9E3779B97F4A7C15ULL);
    return (int)(((h >> 32) * (uint64_t)(uint32_t)count) >> 32);
}

// One row: m distinct indices; with points given, subsets failing the
// fundamental-matrix collinearity check are redrawn (at most 64 times, then the
// row is marked invalid by idx[0] = -1).
void fixed_row(int count, int m, uint32_t row, const float* p1, const float* p2, int* idx)
{
    for (uint32_t u = 0; u < 64; u++) {
        for (int i = 0; i < m; i++) {
            for (uint32_t t = 0;; t++) {
                int v = fixed_draw(count, row, u, (uint32_t)i, t);
                int j = 0;
                for (; j < i; j++)
                    if (idx[j] == v) break;
                if (j == i) { idx[i] = v; break; }
            }
        }
        if (p1 && (have_collinear(p1, idx, m) || have_collinear(p2, idx, m))) continue;
        return;
    }
    idx[0] = -1;
}

int cv_round(double v) { return (int)std::nearbyint(v); }

int update_num_iters(double p, double ep, int model_points, int max_iters)
{
    p = std::max(p, 0.); p = std::min(p, 1.);
    ep = std::max(ep, 0.); ep = std::min(ep, 1.);
    double num = std::max(1. - p, DBL_MIN);
    double denom = 1. - std::pow(1. - ep, model_points);
    if (denom < DBL_MIN) return 0;
    num = std::log(num);
    denom = std::log(denom);
    return denom >= 0 || -num >= max_iters * (-denom) ? max_iters : cv_round(num / denom);
}

void sampson(const double* E, const double* x1, const double* x2, int n, float* err)
{
    for (int i = 0; i < n; i++) {
        double u1 = x1[2 * i], v1 = x1[2 * i + 1], u2 = x2[2 * i], v2 = x2[2 * i + 1];
        double Ex1_0 = (E[0] * u1 + E[1] * v1) + E[2];
        double Ex1_1 = (E[3] * u1 + E[4] * v1) + E[5];
        double Ex1_2 = (E[6] * u1 + E[7] * v1) + E[8];
        double Etx2_0 = (E[0] * u2 + E[3] * v2) + E[6];
        double Etx2_1 = (E[1] * u2 + E[4] * v2) + E[7];
        double s = (u2 * Ex1_0 + v2 * Ex1_1) + Ex1_2;
        double a = Ex1_0 * Ex1_0;
        double b = Ex1_1 * Ex1_1;
        double c = Etx2_0 * Etx2_0;
        double d = Etx2_1 * Etx2_1;
        err[i] = (float)(s * s / (((a + b) + c) + d));
    }
}

void symepi(const double* F, const float* m1, const float* m2, int n, float* err)
{
    for (int i = 0; i < n; i++) {
        double x1 = m1[2 * i], y1 = m1[2 * i + 1], x2 = m2[2 * i], y2 = m2[2 * i + 1];
        double a = (F[0] * x1 + F[1] * y1) + F[2];
        double b = (F[3] * x1 + F[4] * y1) + F[5];
        double c = (F[6] * x1 + F[7] * y1) + F[8];
        double s2 = 1. / (a * a + b * b);
        double d2 = (x2 * a + y2 * b) + c;
        a = (F[0] * x2 + F[3] * y2) + F[6];
        b = (F[1] * x2 + F[4] * y2) + F[7];
        c = (F[2] * x2 + F[5] * y2) + F[8];
        double s1 = 1. / (a * a + b * b);
        double d1 = (x1 * a + y1 * b) + c;
        err[i] = (float)std::max(d1 * d1 * s1, d2 * d2 * s2);
    }
}

// ---- pose recovery -------------------------------------------------------
// cv::recoverPose (called at /root/reference/main.cpp:420 and
// /root/reference/bundle_adjustment.h:80) = cv::decomposeEssentialMat + a
// cheirality vote of the four (R, t) candidates over the correspondences.
// OpenCV's arithmetic (cv::SVD of E, one 4x4 SVD per triangulated point) is
// not restated operation for operation; the restated semantics are: the same
// four candidates in the same order, a point votes for a candidate when its
// triangulated depth lies in (0, distance_thresh) in both cameras, the
// candidate with the most votes wins with OpenCV's tie order.  The singular
// vectors come from a one-sided Jacobi iteration, depths from the two-ray
// least-squares (midpoint) triangulation.  PARITY UNPINNED like the rest.

// A = U diag(S) V^T for a 3x3 row-major A, singular values descending, the
// third left vector taken as u1 x u2 (so det U = +1 and a rank-2 input is
// handled; S[2] is negative when that flips the vector).  Returns 0 when A has
// rank < 2.
int svd3_jacobi(const double* A, double* U, double* S, double* V)
{
    double G[9], W[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    for (int i = 0; i < 9; i++) G[i] = A[i];
    static const int PAIRS[3][2] = {{0, 1}, {0, 2}, {1, 2}};
    for (int sweep = 0; sweep < 30; sweep++) {
        bool rotated = false;
        for (int pi = 0; pi < 3; pi++) {
            const int p = PAIRS[pi][0], q = PAIRS[pi][1];
            const double alpha = (G[p] * G[p] + G[3 + p] * G[3 + p]) + G[6 + p] * G[6 + p];
            const double beta = (G[q] * G[q] + G[3 + q] * G[3 + q]) + G[6 + q] * G[6 + q];
            const double gamma = (G[p] * G[q] + G[3 + p] * G[3 + q]) + G[6 + p] * G[6 + q];
            if (!(std::fabs(gamma) > 0x1p-52 * std::sqrt(alpha * beta))) continue;
            rotated = true;
            const double zeta = (beta - alpha) / (2.0 * gamma);
            const double tt = (zeta >= 0 ? 1.0 : -1.0) / (std::fabs(zeta) + std::sqrt(1.0 + zeta * zeta));
            const double c = 1.0 / std::sqrt(1.0 + tt * tt);
            const double s = c * tt;
            for (int i = 0; i < 3; i++) {
                const double gp = G[3 * i + p], gq = G[3 * i + q];
                G[3 * i + p] = c * gp - s * gq;
                G[3 * i + q] = s * gp + c * gq;
                const double wp = W[3 * i + p], wq = W[3 * i + q];
                W[3 * i + p] = c * wp - s * wq;
                W[3 * i + q] = s * wp + c * wq;
            }
        }
        if (!rotated) break;
    }
    double sig[3];
    for (int j = 0; j < 3; j++)
        sig[j] = std::sqrt((G[j] * G[j] + G[3 + j] * G[3 + j]) + G[6 + j] * G[6 + j]);
    int ord[3] = {0, 1, 2};
    // stable descending order of three values
    if (sig[ord[1]] > sig[ord[0]]) std::swap(ord[0], ord[1]);
    if (sig[ord[2]] > sig[ord[1]]) std::swap(ord[1], ord[2]);
    if (sig[ord[1]] > sig[ord[0]]) std::swap(ord[0], ord[1]);
    if (!(sig[ord[1]] > 0x1p-40 * sig[ord[0]])) return 0;  // numerically rank < 2
    for (int j = 0; j < 3; j++) {
        S[j] = sig[ord[j]];
        for (int i = 0; i < 3; i++) V[3 * i + j] = W[3 * i + ord[j]];
    }
    for (int j = 0; j < 2; j++)
        for (int i = 0; i < 3; i++) U[3 * i + j] = G[3 * i + ord[j]] / S[j];
    U[2] = U[3] * U[7] - U[6] * U[4];
    U[5] = U[6] * U[1] - U[0] * U[7];
    U[8] = U[0] * U[4] - U[3] * U[1];
    // signed, so that the factorisation holds with det U = +1
    S[2] = (U[2] * G[ord[2]] + U[5] * G[3 + ord[2]]) + U[8] * G[6 + ord[2]];
    return 1;
}

double det3(const double* M)
{
    return (M[0] * (M[4] * M[8] - M[5] * M[7]) - M[1] * (M[3] * M[8] - M[5] * M[6])) +
           M[2] * (M[3] * M[7] - M[4] * M[6]);
}

// cv::decomposeEssentialMat: R1 = U W V^T, R2 = U W^T V^T, t = U[:,2] with
// W = [0 1 0; -1 0 0; 0 0 1], after making det U and det V positive.
int decompose_essential(const double* E, double* R1, double* R2, double* t)
{
    double U[9], S[3], V[9];
    if (!svd3_jacobi(E, U, S, V)) return 0;
    if (det3(V) < 0)
        for (int i = 0; i < 9; i++) V[i] = -V[i];
    for (int i = 0; i < 3; i++) {
        for (int j = 0; j < 3; j++) {
            const double a = U[3 * i + 0] * V[3 * j + 1] - U[3 * i + 1] * V[3 * j + 0];
            const double b = U[3 * i + 2] * V[3 * j + 2];
            R1[3 * i + j] = a + b;
            R2[3 * i + j] = b - a;
        }
        t[i] = U[3 * i + 2];
    }
    return 1;
}

// 1 when the two rays meet (in the least-squares sense) at depths inside
// (0, dist) in both cameras.  a = R (x1, y1, 1), b = (x2, y2, 1); minimising
// |z1 a + t - z2 b|^2 gives z1 = n1 / det, z2 = n2 / det with det >= 0.
int cheirality_vote(const double* R, const double* t, double x1, double y1, double x2, double y2,
                    double dist)
{
    const double a0 = (R[0] * x1 + R[1] * y1) + R[2];
    const double a1 = (R[3] * x1 + R[4] * y1) + R[5];
    const double a2 = (R[6] * x1 + R[7] * y1) + R[8];
    const double aa = (a0 * a0 + a1 * a1) + a2 * a2;
    const double bb = (x2 * x2 + y2 * y2) + 1.0;
    const double ab = (a0 * x2 + a1 * y2) + a2;
    const double at = (a0 * t[0] + a1 * t[1]) + a2 * t[2];
    const double bt = (x2 * t[0] + y2 * t[1]) + t[2];
    const double det = aa * bb - ab * ab;
    const double n1 = ab * bt - at * bb;
    const double n2 = aa * bt - ab * at;
    const double lim = dist * det;
    return det > 0 && n1 > 0 && n1 < lim && n2 > 0 && n2 < lim;
}

}  // namespace

extern "C" {

int oracle_track_join(const oracle_dmatch* ab, int n_ab, const oracle_dmatch* bc, int n_bc, int32_t* out,
                      int capacity)
{
    // /root/reference/main.cpp:442-456, loop for loop: every (m, n) with
    // matches[i][m].trainIdx == matches[i+1][n].queryIdx, m outer, n inner.
    if (n_ab < 0 || n_bc < 0 || (n_ab > 0 && !ab) || (n_bc > 0 && !bc)) return -1;
    int count = 0;
    for (int m = 0; m < n_ab; m++)
        for (int n = 0; n < n_bc; n++)
            if (ab[m].trainIdx == bc[n].queryIdx) {
                if (out && count < capacity) {
                    out[3 * count + 0] = ab[m].queryIdx;
                    out[3 * count + 1] = ab[m].trainIdx;
                    out[3 * count + 2] = bc[n].trainIdx;
                }
                count++;
            }
    return count;
}

int oracle_svd3(const double* A, double* U, double* S, double* V) { return svd3_jacobi(A, U, S, V); }

int oracle_decompose_essential(const double* E, double* R1, double* R2, double* t)
{
    return decompose_essential(E, R1, R2, t);
}

int oracle_recover_pose(const double* E, const double* pts1, const double* pts2, int n,
                        const double* K, double distance_thresh, double* R_out, double* t_out,
                        uint8_t* mask, int32_t* votes)
{
    if (!E || !K || n < 0 || (n > 0 && (!pts1 || !pts2)) || !R_out || !t_out) return -1;
    double R1[9], R2[9], t[3], tneg[3];
    if (!decompose_essential(E, R1, R2, t)) return -1;
    for (int i = 0; i < 3; i++) tneg[i] = -t[i];
    const double* Rc[4] = {R1, R2, R1, R2};
    const double* tc[4] = {t, t, tneg, tneg};
    const double fx = K[0], fy = K[4], cx = K[2], cy = K[5];
    std::vector<uint8_t> flags((size_t)n);
    int good[4] = {0, 0, 0, 0};
    for (int i = 0; i < n; i++) {
        uint8_t f = 0;
        if (!mask || mask[i]) {
            const double x1 = (pts1[2 * i] - cx) / fx, y1 = (pts1[2 * i + 1] - cy) / fy;
            const double x2 = (pts2[2 * i] - cx) / fx, y2 = (pts2[2 * i + 1] - cy) / fy;
            for (int k = 0; k < 4; k++)
                if (cheirality_vote(Rc[k], tc[k], x1, y1, x2, y2, distance_thresh)) {
                    f |= (uint8_t)(1 << k);
                    good[k]++;
                }
        }
        flags[i] = f;
    }
    // OpenCV's selection order: the first candidate that is >= all others.
    int best = 3;
    for (int k = 0; k < 4; k++) {
        bool top = true;
        for (int j = 0; j < 4; j++) top = top && good[k] >= good[j];
        if (top) { best = k; break; }
    }
    for (int i = 0; i < 9; i++) R_out[i] = Rc[best][i];
    for (int i = 0; i < 3; i++) t_out[i] = tc[best][i];
    if (mask)
        for (int i = 0; i < n; i++) mask[i] = (uint8_t)((flags[i] >> best) & 1);
    if (votes)
        for (int k = 0; k < 4; k++) votes[k] = good[k];
    return good[best];
}


int oracle_knn_match(const float* query, int nq, size_t qstep, const float* train, int nt,
                     size_t tstep, int dim, int k, oracle_dmatch* out, int32_t* counts,
                     int nthreads)
{
    if (nq < 0 || nt < 0 || dim <= 0 || k <= 0) return -1;
    if (nq == 0) return 0;
    if (nthreads < 1) nthreads = 1;
    nthreads = std::min(nthreads, nq);
    if (nthreads == 1) {
        knn_rows(query, 0, nq, qstep, train, nt, tstep, dim, k, out, counts);
        return 0;
    }
    std::vector<std::thread> th;
    int per = (nq + nthreads - 1) / nthreads;
    for (int t = 0; t < nthreads; t++) {
        int q0 = t * per, q1 = std::min(nq, q0 + per);
        if (q0 >= q1) break;
        th.emplace_back(knn_rows, query, q0, q1, qstep, train, nt, tstep, dim, k, out, counts);
    }
    for (auto& t : th) t.join();
    return 0;
}

int oracle_knn_match_u8path(const float* query, int nq, size_t qstep, const float* train, int nt,
                            size_t tstep, int dim, int k, oracle_dmatch* out, int32_t* counts,
                            int nthreads)
{
    if (nq < 0 || nt < 0 || dim <= 0 || k <= 0) return -1;
    if (nq == 0) return 0;
    std::vector<uint8_t> qb, tb;
    if (!narrow_rows(query, nq, qstep, dim, qb) || !narrow_rows(train, nt, tstep, dim, tb) ||
        (double)dim * 255.0 * 255.0 >= 16777216.0)
        return oracle_knn_match(query, nq, qstep, train, nt, tstep, dim, k, out, counts, nthreads);
    if (nthreads < 1) nthreads = 1;
    // blocks of four query rows, dealt in contiguous runs
    const int blocks = (nq + 3) / 4;
    nthreads = std::min(nthreads, blocks);
    std::vector<std::thread> th;
    const int per = (blocks + nthreads - 1) / nthreads * 4;
    for (int t = 0; t < nthreads; t++) {
        const int q0 = t * per, q1 = std::min(nq, q0 + per);
        if (q0 >= q1) break;
        if (nthreads == 1) knn_rows_u8(qb.data(), q0, q1, tb.data(), nt, dim, k, out, counts);
        else th.emplace_back(knn_rows_u8, qb.data(), q0, q1, tb.data(), nt, dim, k, out, counts);
    }
    for (auto& t : th) t.join();
    return 0;
}

int oracle_ratio_test(const oracle_dmatch* matches, const int32_t* counts, int nq, int k,
                      double ratio, int32_t* keep_rows)
{
    int n = 0;
    for (int r = 0; r < nq; r++) {
        if (counts[r] < 2) continue;
        const oracle_dmatch* m = matches + (size_t)r * k;
        if ((double)m[0].distance < ratio * (double)m[1].distance) keep_rows[n++] = r;
    }
    return n;
}

void oracle_rng_next(uint64_t seed, int n, uint32_t* out)
{
    CvRng rng(seed);
    for (int i = 0; i < n; i++) out[i] = rng.next();
}

void oracle_rng_uniform(uint64_t seed, int a, int b, int n, int32_t* out)
{
    CvRng rng(seed);
    for (int i = 0; i < n; i++) out[i] = rng.uniform(a, b);
}

int oracle_sample_table(int count, int m, int n_iters, int32_t* out)
{
    if (count < m || m <= 0) return -1;
    CvRng rng((uint64_t)-1);
    for (int it = 0; it < n_iters; it++)
        if (!get_subset(rng, count, m, out + (size_t)it * m, nullptr, nullptr, 10000)) return -1;
    return 0;
}

int oracle_sample_table_f(int count, const float* pts1, const float* pts2, int n_iters,
                          int32_t* out)
{
    if (count < 7) return -1;
    CvRng rng((uint64_t)-1);
    for (int it = 0; it < n_iters; it++)
        if (!get_subset(rng, count, 7, out + (size_t)it * 7, pts1, pts2, 10000)) return it;
    return n_iters;
}

int oracle_sample_table_fixed(int count, int m, int n_rows, const float* pts1, const float* pts2,
                              int32_t* out)
{
    if (count <= m || m <= 0 || m > 7) return -1;
    for (int r = 0; r < n_rows; r++) fixed_row(count, m, (uint32_t)r, pts1, pts2, out + (size_t)r * m);
    return 0;
}

int oracle_ransac_update_num_iters(double p, double ep, int model_points, int max_iters)
{
    return update_num_iters(p, ep, model_points, max_iters);
}

int oracle_real_roots(const double* coef, int degree, double* roots)
{
    if (degree < 1 || degree > 10) return -1;
    return real_roots(coef, degree, roots);
}

int oracle_five_point(const double* x1, const double* x2, double* models)
{
    return five_point(x1, x2, models);
}

int oracle_seven_point(const float* x1, const float* x2, double* models)
{
    return seven_point(x1, x2, models);
}

void oracle_sampson_errors(const double* E, const double* x1, const double* x2, int n, float* err)
{
    sampson(E, x1, x2, n, err);
}

void oracle_symepi_errors(const double* F, const float* x1, const float* x2, int n, float* err)
{
    symepi(F, x1, x2, n, err);
}

double oracle_normalize_points(const double* pts_px, int n, const double* K, double threshold_px,
                               double* out)
{
    double fx = K[0], fy = K[4], cx = K[2], cy = K[5];
    for (int i = 0; i < n; i++) {
        out[2 * i] = (pts_px[2 * i] - cx) / fx;
        out[2 * i + 1] = (pts_px[2 * i + 1] - cy) / fy;
    }
    return threshold_px / ((fx + fy) / 2);
}

// Fixed-count RANSAC with the hypotheses spread over threads (the timed CPU
// baseline of the benchmark; the result is the sequential loop's: the first
// hypothesis and model, in scan order, that reaches the largest inlier count).
int oracle_ransac_fixed_mt(int model_kind, const double* pts1, const double* pts2, int n, const double* K,
                           double threshold_px, int fixed_count, int nthreads, uint8_t* mask, double* model_out,
                           int32_t* n_inliers)
{
    if ((model_kind != 0 && model_kind != 1) || n < 0 || fixed_count < 1) return -1;
    const int m = model_kind == 0 ? 5 : 7;
    if (n_inliers) *n_inliers = 0;
    if (n <= m) return oracle_ransac(model_kind, pts1, pts2, n, K, 0.99, threshold_px, 1000, fixed_count, nullptr, mask,
                                     model_out, n_inliers, nullptr, nullptr);
    std::vector<double> x1, x2;
    std::vector<float> f1, f2;
    double thr = threshold_px;
    if (model_kind == 0) {
        x1.resize(2 * (size_t)n);
        x2.resize(2 * (size_t)n);
        thr = oracle_normalize_points(pts1, n, K, threshold_px, x1.data());
        oracle_normalize_points(pts2, n, K, threshold_px, x2.data());
    } else {
        f1.resize(2 * (size_t)n);
        f2.resize(2 * (size_t)n);
        for (int i = 0; i < 2 * n; i++) { f1[i] = (float)pts1[i]; f2[i] = (float)pts2[i]; }
    }
    const float t = (float)(thr * thr);
    if (nthreads < 1) nthreads = 1;
    nthreads = std::min(nthreads, fixed_count);
    struct Best { int good = 0; double model[9]; };
    std::vector<Best> best(nthreads);
    auto work = [&](int tid, int h0, int h1) {
        std::vector<float> err(n);
        Best& b = best[tid];
        double models[10 * 9];
        for (int iter = h0; iter < h1; iter++) {
            int idx[7];
            fixed_row(n, m, (uint32_t)iter, model_kind == 1 ? f1.data() : nullptr, model_kind == 1 ? f2.data() : nullptr,
                      idx);
            if (idx[0] < 0) continue;
            int nm;
            if (model_kind == 0) {
                double a[10], c[10];
                for (int i = 0; i < 5; i++) {
                    a[2 * i] = x1[2 * idx[i]]; a[2 * i + 1] = x1[2 * idx[i] + 1];
                    c[2 * i] = x2[2 * idx[i]]; c[2 * i + 1] = x2[2 * idx[i] + 1];
                }
                nm = five_point(a, c, models);
            } else {
                float a[14], c[14];
                for (int i = 0; i < 7; i++) {
                    a[2 * i] = f1[2 * idx[i]]; a[2 * i + 1] = f1[2 * idx[i] + 1];
                    c[2 * i] = f2[2 * idx[i]]; c[2 * i + 1] = f2[2 * idx[i] + 1];
                }
                nm = seven_point(a, c, models);
            }
            for (int q = 0; q < nm; q++) {
                if (model_kind == 0) sampson(models + 9 * q, x1.data(), x2.data(), n, err.data());
                else symepi(models + 9 * q, f1.data(), f2.data(), n, err.data());
                int good = 0;
                for (int i = 0; i < n; i++) good += err[i] <= t;
                if (good > std::max(b.good, m - 1)) {
                    b.good = good;
                    std::memcpy(b.model, models + 9 * q, sizeof(b.model));
                }
            }
        }
    };
    const int per = (fixed_count + nthreads - 1) / nthreads;
    std::vector<std::thread> th;
    for (int k = 0; k < nthreads; k++) th.emplace_back(work, k, k * per, std::min(fixed_count, (k + 1) * per));
    for (auto& x : th) x.join();
    int win = -1, good = 0;
    for (int k = 0; k < nthreads; k++)   // ascending hypothesis ranges, strict >: the sequential loop's choice
        if (best[k].good > std::max(good, m - 1)) { good = best[k].good; win = k; }
    if (win < 0) return 0;
    std::memcpy(model_out, best[win].model, 9 * sizeof(double));
    std::vector<float> err(n);
    if (model_kind == 0) sampson(model_out, x1.data(), x2.data(), n, err.data());
    else symepi(model_out, f1.data(), f2.data(), n, err.data());
    for (int i = 0; i < n; i++) mask[i] = (uint8_t)(err[i] <= t);
    if (n_inliers) *n_inliers = good;
    return 1;
}

// candidates of the last direct (n == m) solve on this thread
static thread_local int g_direct_n = 0;
static thread_local double g_direct_models[10 * 9];

int oracle_ransac_direct_models(double* models_out, int capacity)
{
    if (g_direct_n > capacity) return -1;
    std::memcpy(models_out, g_direct_models, (size_t)g_direct_n * 9 * sizeof(double));
    return g_direct_n;
}

int oracle_ransac(int model_kind, const double* pts1, const double* pts2, int n, const double* K,
                  double prob, double threshold_px, int max_iters, int fixed_count,
                  const int32_t* sample_table, uint8_t* mask, double* model_out,
                  int32_t* n_inliers, int32_t* iters_used, int32_t* hyp_counts)
{
    if (model_kind != 0 && model_kind != 1) return -1;
    if (n < 0 || !(prob > 0 && prob < 1)) return -1;
    const int m = model_kind == 0 ? 5 : 7;
    if (n_inliers) *n_inliers = 0;
    if (iters_used) *iters_used = 0;
    if (n < m) return 0;

    std::vector<double> x1, x2;
    std::vector<float> f1, f2;
    double thr = threshold_px;
    if (model_kind == 0) {
        x1.resize(2 * (size_t)n);
        x2.resize(2 * (size_t)n);
        thr = oracle_normalize_points(pts1, n, K, threshold_px, x1.data());
        oracle_normalize_points(pts2, n, K, threshold_px, x2.data());
    } else {
        f1.resize(2 * (size_t)n);
        f2.resize(2 * (size_t)n);
        for (int i = 0; i < 2 * n; i++) { f1[i] = (float)pts1[i]; f2[i] = (float)pts2[i]; }
    }
    const float t = (float)(thr * thr);

    auto solve = [&](const int* idx, double* models) -> int {
        if (model_kind == 0) {
            double a[10], b[10];
            for (int i = 0; i < 5; i++) {
                a[2 * i] = x1[2 * idx[i]]; a[2 * i + 1] = x1[2 * idx[i] + 1];
                b[2 * i] = x2[2 * idx[i]]; b[2 * i + 1] = x2[2 * idx[i] + 1];
            }
            return five_point(a, b, models);
        }
        float a[14], b[14];
        for (int i = 0; i < 7; i++) {
            a[2 * i] = f1[2 * idx[i]]; a[2 * i + 1] = f1[2 * idx[i] + 1];
            b[2 * i] = f2[2 * idx[i]]; b[2 * i + 1] = f2[2 * idx[i] + 1];
        }
        return seven_point(a, b, models);
    };
    std::vector<float> err(n);
    auto errors = [&](const double* M) {
        if (model_kind == 0) sampson(M, x1.data(), x2.data(), n, err.data());
        else symepi(M, f1.data(), f2.data(), n, err.data());
    };

    double models[10 * 9];
    if (n == m) {
        int idx[7];
        for (int i = 0; i < m; i++) idx[i] = i;
        int nm = solve(idx, models);
        if (nm <= 0) return 0;
        // OpenCV returns every candidate stacked; model_out takes the first, the
        // whole stack goes to direct_models_out when the caller wants it
        g_direct_n = nm;
        std::memcpy(g_direct_models, models, (size_t)nm * 9 * sizeof(double));
        std::memcpy(model_out, models, 9 * sizeof(double));
        std::memset(mask, 1, n);
        if (n_inliers) *n_inliers = n;
        return 1;
    }

    CvRng rng((uint64_t)-1);
    int niters = fixed_count > 0 ? fixed_count : std::max(max_iters, 1);
    int best = 0;
    std::vector<uint8_t> cur(n);
    double best_model[9] = {0};
    int iter = 0;
    for (; iter < niters; iter++) {
        int idx[7];
        if (sample_table) {
            for (int i = 0; i < m; i++) idx[i] = sample_table[(size_t)iter * m + i];
        } else if (fixed_count > 0) {
            fixed_row(n, m, (uint32_t)iter, model_kind == 1 ? f1.data() : nullptr,
                      model_kind == 1 ? f2.data() : nullptr, idx);
        } else {
            bool found = get_subset(rng, n, m, idx, model_kind == 1 ? f1.data() : nullptr,
                                    model_kind == 1 ? f2.data() : nullptr, 10000);
            if (!found) {
                if (iter == 0) return 0;
                break;
            }
        }
        if (hyp_counts)
            for (int q = 0; q < 10; q++) hyp_counts[(size_t)iter * 10 + q] = -1;
        int nm = idx[0] < 0 ? 0 : solve(idx, models);
        if (nm <= 0) continue;
        for (int q = 0; q < nm; q++) {
            errors(models + 9 * q);
            int good = 0;
            for (int i = 0; i < n; i++) {
                int f = err[i] <= t;
                cur[i] = (uint8_t)f;
                good += f;
            }
            if (hyp_counts) hyp_counts[(size_t)iter * 10 + q] = good;
            if (good > std::max(best, m - 1)) {
                std::memcpy(mask, cur.data(), n);
                std::memcpy(best_model, models + 9 * q, sizeof(best_model));
                best = good;
                if (fixed_count <= 0)
                    niters = update_num_iters(prob, (double)(n - good) / n, m, niters);
            }
        }
    }
    if (iters_used) *iters_used = iter;
    if (best > 0) {
        std::memcpy(model_out, best_model, sizeof(best_model));
        if (n_inliers) *n_inliers = best;
        return 1;
    }
    return 0;
}

}  // extern "C"
