// Micro-benchmark and numerics probe for a matrix-pipe screen of the RANSAC
// scoring test (symmetric epipolar distance, fundamental model).
//
// For a tile of 32 points x 32 models a wave forms
//   d      = sum_k m_k * p_k           9 monomials, v_mfma_f32_32x32x2_f32 x5
//   in_j   = lower bound of T * den_j   6 quadratic monomials each,
//   out_j  = upper bound of T * den_j   v_mfma_f32_32x32x16_f16 x4
// and settles each test with one multiply and four compares:
//   inlier  if d^2 <= in_1  and d^2 <= in_2
//   outlier if d^2 >= out_1 or  d^2 >= out_2
// The program checks (1) that the f32 MFMA equals a k-ordered fmaf chain bit
// for bit, (2) the error of the f16 MFMA against exact sums, subnormal
// operands included, (3) that no proven verdict contradicts the double
// precision verdict, and times the loop against the vector-only filter.
//   hipcc --offload-arch=gfx950 -O3 scripts/micro/mfma_screen.hip -o scripts/micro/mfma_screen
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <random>
#include <vector>

typedef float fx16 __attribute__((ext_vector_type(16)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f2v __attribute__((ext_vector_type(2)));

#define CHECK(x)                                                                  \
    do {                                                                          \
        hipError_t e_ = (x);                                                      \
        if (e_ != hipSuccess) {                                                   \
            printf("%s failed: %s\n", #x, hipGetErrorString(e_));                 \
            return 1;                                                             \
        }                                                                         \
    } while (0)

// ---- the screen ---------------------------------------------------------------
// monoD [tile][10][32] f32   monomials of d, k-major inside a tile
// monoQ [tile][32][16] f16   quadratic monomials (6 of side 1, 6 of side 2, 4 zero)
// Bd    [model][10]   f32    coefficients of d (scaled)
// Bq    [4][model][16] f16   in_1, in_2, out_1, out_2 coefficient columns
template <bool DUMP, int MODE = 0>
__global__ void __launch_bounds__(64, 2) screen_kernel(const float* __restrict__ monoD, const h8* __restrict__ monoQ,
                                                   const float* __restrict__ Bd, const h8* __restrict__ Bq,
                                                   int n_models, int n_tiles, int* __restrict__ cnt_out,
                                                   int* __restrict__ und_out, float* __restrict__ dump)
{
    const int lane = threadIdx.x, r = lane & 31, h = lane >> 5;
    const int model = blockIdx.x * 32 + r;
    float bd[5];
#pragma unroll
    for (int j = 0; j < 5; ++j) bd[j] = Bd[(size_t)model * 10 + 2 * j + h];
    h8 bq[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) bq[t] = Bq[((size_t)t * n_models + model) * 2 + h];
    int cnt = 0, und = 0;
    float ad[5];
    h8 aq;
    // running lane pointers: a tile advances them by constants
    const float* pd = monoD + h * 32 + r;
    const h8* pq = monoQ + r * 2 + h;
    auto fetch = [&]() {
#pragma unroll
        for (int j = 0; j < 5; ++j) ad[j] = pd[j * 64];
        aq = *pq;
    };
    fetch();
    for (int tile = 0; tile < n_tiles; ++tile) {
        float cd[5];
#pragma unroll
        for (int j = 0; j < 5; ++j) cd[j] = ad[j];
        const h8 cq = aq;
        if (tile + 1 < n_tiles) {
            pd += 320;
            pq += 64;
        }
        fetch();
        fx16 d = {0};
        fx16 q[4];
        if (MODE == 2) {
            // no matrix instruction: stand-in values from the loaded operands
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                d[i] = cd[i % 5] + bd[i % 5] * (float)i;
#pragma unroll
                for (int t = 0; t < 4; ++t) q[t][i] = (float)cq[(i + t) & 7] + (float)bq[t][i & 7];
            }
        } else {
            if (MODE == 3) {
#pragma unroll
                for (int j = 0; j < 3; ++j) d = __builtin_amdgcn_mfma_f32_32x32x16_f16(cq, bq[j], d, 0, 0, 0);
                d[0] += cd[0] + cd[1] + cd[2] + cd[3] + cd[4] + bd[0] + bd[1] + bd[2] + bd[3] + bd[4];
            } else {
#pragma unroll
                for (int j = 0; j < 5; ++j) d = __builtin_amdgcn_mfma_f32_32x32x2f32(cd[j], bd[j], d, 0, 0, 0);
            }
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                fx16 z = {0};
                q[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(cq, bq[t], z, 0, 0, 0);
            }
        }
        if (MODE == 1) {
            float m = d[0];
#pragma unroll
            for (int i = 1; i < 16; ++i) m = fminf(m, d[i]);
            float qm = q[0][0];
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int i = 0; i < 16; i += 8) qm = fminf(qm, q[t][i]);
            cnt += m <= qm ? 1 : 0;
            continue;
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const float n = d[i] * d[i];
            const bool in = (n <= q[0][i]) && (n <= q[1][i]);
            const bool out = (n >= q[2][i]) || (n >= q[3][i]);
            cnt += in ? 1 : 0;
            und += (!in && !out) ? 1 : 0;
            if (DUMP) {
                const int row = (i & 3) + 8 * (i >> 2) + 4 * h;
                float* o = dump + (((size_t)tile * 32 + row) * n_models + model) * 5;
                o[0] = d[i];
                o[1] = q[0][i];
                o[2] = q[1][i];
                o[3] = q[2][i];
                o[4] = q[3][i];
            }
        }
    }
    atomicAdd(&cnt_out[model], cnt);
    atomicAdd(&und_out[model], und);
}

// ---- the vector-only filter, shaped like the shipped scoring loop ---------------
struct VConst {
    float c1, c2, c3, c4, kin, kout;
};
__device__ __forceinline__ f2v fma2(f2v a, f2v b, f2v c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ f2v splat(float v) { return (f2v)(v); }
__device__ __forceinline__ f2v min2(f2v a, f2v b) { return f2v{fminf(a.x, b.x), fminf(a.y, b.y)}; }
__global__ void __launch_bounds__(64) vector_kernel(const float* __restrict__ soa, const float* __restrict__ Mf_all,
                                                   const VConst* __restrict__ vc, int n_pts, int* __restrict__ cnt_out,
                                                   int* __restrict__ und_out)
{
    typedef const f2v __attribute__((address_space(4))) cf2v;
    const int model = blockIdx.x * 64 + threadIdx.x;
    float Mf[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) Mf[k] = Mf_all[(size_t)model * 9 + k];
    const VConst fc = vc[model];
    int cnt = 0, und = 0;
    for (int i = 0; i < n_pts; i += 4) {
        cf2v* B = (cf2v*)(uintptr_t)(soa + (size_t)(i >> 2) * 16);
        f2v cur[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) cur[k] = B[k];
        f2v hin[2], hout[2];
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const f2v x1 = cur[s], y1 = cur[2 + s], x2 = cur[4 + s], y2 = cur[6 + s];
            const f2v a = fma2(splat(Mf[0]), x1, fma2(splat(Mf[1]), y1, splat(Mf[2])));
            const f2v b = fma2(splat(Mf[3]), x1, fma2(splat(Mf[4]), y1, splat(Mf[5])));
            const f2v c = fma2(splat(Mf[6]), x1, fma2(splat(Mf[7]), y1, splat(Mf[8])));
            const f2v d = fma2(x2, a, fma2(y2, b, c));
            const f2v at = fma2(splat(Mf[0]), x2, fma2(splat(Mf[3]), y2, splat(Mf[6])));
            const f2v bt = fma2(splat(Mf[1]), x2, fma2(splat(Mf[4]), y2, splat(Mf[7])));
            const f2v n = d * d;
            const f2v den = min2(fma2(a, a, b * b), fma2(at, at, bt * bt));
            hin[s] = fma2(n, splat(fc.c1), fma2(den, splat(-fc.c2), splat(fc.kin)));
            hout[s] = fma2(n, splat(fc.c3), fma2(den, splat(-fc.c4), splat(-fc.kout)));
        }
        const bool in0 = hin[0].x <= 0.f, in1 = hin[0].y <= 0.f, in2 = hin[1].x <= 0.f, in3 = hin[1].y <= 0.f;
        cnt += (in0 ? 1 : 0) + (in1 ? 1 : 0) + (in2 ? 1 : 0) + (in3 ? 1 : 0);
        und += (!(in0 || hout[0].x >= 0.f) ? 1 : 0) + (!(in1 || hout[0].y >= 0.f) ? 1 : 0) +
               (!(in2 || hout[1].x >= 0.f) ? 1 : 0) + (!(in3 || hout[1].y >= 0.f) ? 1 : 0);
    }
    atomicAdd(&cnt_out[model], cnt);
    atomicAdd(&und_out[model], und);
}

// ---- host ---------------------------------------------------------------------
static float up(double v) { return (float)(v * (1.0 + 1.0 / 1048576.0)); }
static float down(double v) { return (float)(v * (1.0 - 1.0 / 1048576.0)); }

struct Packed {
    float bd[10];
    _Float16 bq[4][16];
    float Mf[9];
    VConst vc;
};

// Coefficient columns of one model.  R = max |coordinate|, tf = threshold^2.
static Packed pack_model(const double* m, double R, float tf)
{
    Packed P;
    memset(&P, 0, sizeof P);
    const double u = 1.0 / 16777216.0, eps = 1.0 / 512.0;
    const double Tin = (double)tf * (1.0 - 1.0 / 65536.0), Tout = (double)tf * (1.0 + 1.0 / 65536.0);
    double am[9];
    for (int i = 0; i < 9; ++i) am[i] = fabs(m[i]);
    // bound of sum |m_k p_k| over the coordinate box, both groupings
    const double r0 = (am[0] + am[1]) * R + am[2], r1 = (am[3] + am[4]) * R + am[5], r2 = (am[6] + am[7]) * R + am[8];
    const double S = R * (r0 + r1) + r2;
    const double dd = 16 * u * S;                   // |d~ - d*|
    const double kn = dd * dd * (1.0 + 1.0 / eps);  // d*^2 within d~^2 (1 -+ eps) -+ kn
    // quadratic forms: side A rows (m0 m1 m2),(m3 m4 m5) in (x1,y1); side B columns in (x2,y2)
    double c[2][6];
    const double a0 = m[0], a1 = m[1], a2 = m[2], b0 = m[3], b1 = m[4], b2 = m[5];
    c[0][0] = (a0 * a0 + b0 * b0) * R * R;
    c[0][1] = 2 * (a0 * a1 + b0 * b1) * R * R;
    c[0][2] = (a1 * a1 + b1 * b1) * R * R;
    c[0][3] = 2 * (a0 * a2 + b0 * b2) * R;
    c[0][4] = 2 * (a1 * a2 + b1 * b2) * R;
    c[0][5] = a2 * a2 + b2 * b2;
    const double p0 = m[0], p1 = m[3], p2 = m[6], q0 = m[1], q1 = m[4], q2 = m[7];
    c[1][0] = (p0 * p0 + q0 * q0) * R * R;
    c[1][1] = 2 * (p0 * p1 + q0 * q1) * R * R;
    c[1][2] = (p1 * p1 + q1 * q1) * R * R;
    c[1][3] = 2 * (p0 * p2 + q0 * q2) * R;
    c[1][4] = 2 * (p1 * p2 + q1 * q2) * R;
    c[1][5] = p2 * p2 + q2 * q2;
    const double delta = 1.0 / 512.0;  // relative error budget of the f16 product sum
    double w[4][6];
    double wmax = 0;
    for (int s = 0; s < 2; ++s) {
        // sum |c_k q_k| <= e0 x^2 + e2 y^2 + e5 with |xy| <= (x^2+y^2)/2, |x| <= (x^2+1)/2
        const double e0 = fabs(c[s][0]) + 0.5 * fabs(c[s][1]) + 0.5 * fabs(c[s][3]);
        const double e2 = fabs(c[s][2]) + 0.5 * fabs(c[s][1]) + 0.5 * fabs(c[s][4]);
        const double e5 = fabs(c[s][5]) + 0.5 * fabs(c[s][3]) + 0.5 * fabs(c[s][4]);
        const double fin = Tin / ((1.0 + eps) * (1.0 + 4 * u)), fout = Tout / ((1.0 - eps) * (1.0 - 4 * u));
        for (int k = 0; k < 6; ++k) {
            w[s][k] = c[s][k] * fin;
            w[2 + s][k] = c[s][k] * fout;
        }
        w[s][0] -= delta * e0 * fin;
        w[s][2] -= delta * e2 * fin;
        w[s][5] -= delta * e5 * fin + kn / ((1.0 + eps) * (1.0 + 4 * u));
        w[2 + s][0] += delta * e0 * fout;
        w[2 + s][2] += delta * e2 * fout;
        w[2 + s][5] += delta * e5 * fout + kn / ((1.0 - eps) * (1.0 - 4 * u));
    }
    for (int t = 0; t < 4; ++t)
        for (int k = 0; k < 6; ++k) wmax = fmax(wmax, fabs(w[t][k]));
    // scale: a power of four bringing the largest coefficient into [2^9, 2^11)
    int e;
    frexp(wmax, &e);  // wmax = f * 2^e, f in [0.5,1)
    int sh = 10 - e;
    if (sh & 1) sh += 1;
    const double s2 = ldexp(1.0, sh), s1 = ldexp(1.0, sh / 2);
    const double flush = 12.0 / 16384.0;  // operands below the f16 normal range may vanish
    for (int t = 0; t < 4; ++t) {
        const int side = t & 1;
        for (int k = 0; k < 6; ++k) {
            double v = w[t][k] * s2;
            if (k == 5) v += (t < 2 ? -flush : flush);
            P.bq[t][side * 6 + k] = (_Float16)v;
        }
    }
    for (int k = 0; k < 9; ++k) P.bd[k] = (float)(m[k] * s1);
    for (int k = 0; k < 9; ++k) P.Mf[k] = (float)m[k];
    // the vector filter's constants (same structure as the shipped make_filter)
    {
        const double g = 1.0 + 1.0 / 262144.0, ke = (1.0 + 1.0 / eps) * (1.0 + 1.0 / 1024.0);
        const double qq0 = (am[0] + am[3]) * R + am[6], qq1 = (am[1] + am[4]) * R + am[7];
        const double kd = fmax((8 * u * qq0) * (8 * u * qq0) + (8 * u * qq1) * (8 * u * qq1),
                               (8 * u * r0) * (8 * u * r0) + (8 * u * r1) * (8 * u * r1)) * ke;
        const double knv = (16 * u * S) * (16 * u * S) * ke;
        P.vc.c1 = up((1.0 + eps) * g);
        P.vc.c2 = down((1.0 - eps) * Tin / g);
        P.vc.c3 = down((1.0 - eps) / g);
        P.vc.c4 = up((1.0 + eps) * Tout * g);
        P.vc.kin = up((knv + kd * Tin) * g);
        P.vc.kout = up((knv + kd * Tout) * g);
    }
    return P;
}

static int exact_inlier(const double* m, double x1, double y1, double x2, double y2, double t)
{
    const double a = m[0] * x1 + m[1] * y1 + m[2], b = m[3] * x1 + m[4] * y1 + m[5], c = m[6] * x1 + m[7] * y1 + m[8];
    const double d = x2 * a + y2 * b + c;
    const double at = m[0] * x2 + m[3] * y2 + m[6], bt = m[1] * x2 + m[4] * y2 + m[7];
    const double r = fmax(d * d / (a * a + b * b), d * d / (at * at + bt * bt));
    return r <= t ? 1 : 0;
}

int main()
{
    const int n_pts = 3200, n_tiles = n_pts / 32;
    const int n_models = 32 * 1024 * 8;  // 8 waves per SIMD in one launch at most
    const int n_check_models = 512;
    const float tf = 1.0f;
    std::mt19937_64 rng(7);
    std::uniform_real_distribution<double> U(0, 1);
    std::normal_distribution<double> N(0, 1);
    // a planted fundamental matrix from two cameras
    double Ft[9];
    {
        const double f = 2759.0, cx = 2000, cy = 1500;
        const double ang = 0.05, t[3] = {1.0, 0.1, 0.2};
        const double Rm[9] = {cos(ang), 0, sin(ang), 0, 1, 0, -sin(ang), 0, cos(ang)};
        const double tx[9] = {0, -t[2], t[1], t[2], 0, -t[0], -t[1], t[0], 0};
        double E[9];
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) {
                E[3 * i + j] = 0;
                for (int k = 0; k < 3; ++k) E[3 * i + j] += tx[3 * i + k] * Rm[3 * k + j];
            }
        const double Ki[9] = {1 / f, 0, -cx / f, 0, 1 / f, -cy / f, 0, 0, 1};
        double T1[9];
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) {
                T1[3 * i + j] = 0;
                for (int k = 0; k < 3; ++k) T1[3 * i + j] += Ki[3 * k + i] * E[3 * k + j];  // Ki^T E
            }
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) {
                Ft[3 * i + j] = 0;
                for (int k = 0; k < 3; ++k) Ft[3 * i + j] += T1[3 * i + k] * Ki[3 * k + j];
            }
    }
    std::vector<double> px(4 * n_pts);
    double R = 0;
    for (int i = 0; i < n_pts; ++i) {
        const double x1 = (float)(U(rng) * 4000), y1 = (float)(U(rng) * 3000);
        double x2, y2;
        if (U(rng) < 0.93) {
            // a point on the epipolar line of (x1,y1) in image 2, plus noise
            const double a = Ft[0] * x1 + Ft[1] * y1 + Ft[2], b = Ft[3] * x1 + Ft[4] * y1 + Ft[5],
                         c = Ft[6] * x1 + Ft[7] * y1 + Ft[8];
            x2 = U(rng) * 4000;
            y2 = -(a * x2 + c) / b;
            const double nn = sqrt(a * a + b * b);
            const double off = N(rng) * 0.6;
            x2 += off * a / nn;
            y2 += off * b / nn;
        } else {
            x2 = U(rng) * 4000;
            y2 = U(rng) * 3000;
        }
        x2 = (float)x2;
        y2 = (float)y2;
        px[4 * i] = x1;
        px[4 * i + 1] = y1;
        px[4 * i + 2] = x2;
        px[4 * i + 3] = y2;
        for (int k = 0; k < 4; ++k) R = fmax(R, fabs(px[4 * i + k]));
    }
    // models: 12 % near the planted one, the rest random
    std::vector<double> models((size_t)n_models * 9);
    for (int j = 0; j < n_models; ++j) {
        const bool good = U(rng) < 0.12;
        for (int k = 0; k < 9; ++k) {
            if (good) models[(size_t)j * 9 + k] = Ft[k] * (1.0 + 2e-4 * N(rng));
            else {
                const double scale = (k == 8) ? 1.0 : (k == 2 || k == 5 || k == 6 || k == 7) ? 1e-3 : 1e-6;
                models[(size_t)j * 9 + k] = scale * N(rng);
            }
        }
    }
    // operands
    std::vector<float> monoD((size_t)n_tiles * 10 * 32, 0.f), soa((size_t)n_pts * 4);
    std::vector<_Float16> monoQ((size_t)n_tiles * 32 * 16, (_Float16)0.f);
    for (int i = 0; i < n_pts; ++i) {
        const double x1 = px[4 * i], y1 = px[4 * i + 1], x2 = px[4 * i + 2], y2 = px[4 * i + 3];
        const double mono[9] = {x2 * x1, x2 * y1, x2, y2 * x1, y2 * y1, y2, x1, y1, 1.0};
        const int tile = i >> 5, r = i & 31;
        for (int k = 0; k < 9; ++k) monoD[((size_t)tile * 10 + k) * 32 + r] = (float)mono[k];
        const double s1x = x1 / R, s1y = y1 / R, s2x = x2 / R, s2y = y2 / R;
        const double q[12] = {s1x * s1x, s1x * s1y, s1y * s1y, s1x, s1y, 1.0, s2x * s2x, s2x * s2y, s2y * s2y, s2x, s2y, 1.0};
        for (int k = 0; k < 12; ++k) monoQ[((size_t)tile * 32 + r) * 16 + k] = (_Float16)q[k];
        float* blk = &soa[(size_t)(i >> 2) * 16];
        blk[0 + (i & 3)] = (float)x1;
        blk[4 + (i & 3)] = (float)y1;
        blk[8 + (i & 3)] = (float)x2;
        blk[12 + (i & 3)] = (float)y2;
    }
    std::vector<float> Bd((size_t)n_models * 10, 0.f), Mf((size_t)n_models * 9);
    std::vector<_Float16> Bq((size_t)4 * n_models * 16, (_Float16)0.f);
    std::vector<VConst> vc(n_models);
    for (int j = 0; j < n_models; ++j) {
        const Packed P = pack_model(&models[(size_t)j * 9], R, tf);
        for (int k = 0; k < 10; ++k) Bd[(size_t)j * 10 + k] = P.bd[k];
        for (int t = 0; t < 4; ++t)
            for (int k = 0; k < 16; ++k) Bq[((size_t)t * n_models + j) * 16 + k] = P.bq[t][k];
        for (int k = 0; k < 9; ++k) Mf[(size_t)j * 9 + k] = P.Mf[k];
        vc[j] = P.vc;
    }
    float *d_monoD, *d_Bd, *d_soa, *d_Mf, *d_dump;
    h8 *d_monoQ, *d_Bq;
    VConst* d_vc;
    int *d_cnt, *d_und;
    CHECK(hipMalloc(&d_monoD, monoD.size() * 4));
    CHECK(hipMalloc(&d_monoQ, monoQ.size() * 2));
    CHECK(hipMalloc(&d_Bd, Bd.size() * 4));
    CHECK(hipMalloc(&d_Bq, Bq.size() * 2));
    CHECK(hipMalloc(&d_soa, soa.size() * 4));
    CHECK(hipMalloc(&d_Mf, Mf.size() * 4));
    CHECK(hipMalloc(&d_vc, vc.size() * sizeof(VConst)));
    CHECK(hipMalloc(&d_cnt, n_models * 4));
    CHECK(hipMalloc(&d_und, n_models * 4));
    CHECK(hipMalloc(&d_dump, (size_t)n_pts * n_check_models * 5 * 4));
    CHECK(hipMemcpy(d_monoD, monoD.data(), monoD.size() * 4, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(d_monoQ, monoQ.data(), monoQ.size() * 2, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(d_Bd, Bd.data(), Bd.size() * 4, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(d_soa, soa.data(), soa.size() * 4, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(d_Mf, Mf.data(), Mf.size() * 4, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(d_vc, vc.data(), vc.size() * sizeof(VConst), hipMemcpyHostToDevice));

    // ---- numerics on the first n_check_models models ----
    {
        // a compact copy of Bq for the checked models ([4][n_check][16])
        std::vector<_Float16> Bqc((size_t)4 * n_check_models * 16);
        for (int t = 0; t < 4; ++t)
            for (int j = 0; j < n_check_models; ++j)
                for (int k = 0; k < 16; ++k)
                    Bqc[((size_t)t * n_check_models + j) * 16 + k] = Bq[((size_t)t * n_models + j) * 16 + k];
        CHECK(hipMemcpy(d_Bq, Bqc.data(), Bqc.size() * 2, hipMemcpyHostToDevice));
        CHECK(hipMemset(d_cnt, 0, n_models * 4));
        CHECK(hipMemset(d_und, 0, n_models * 4));
        screen_kernel<true><<<n_check_models / 32, 64>>>(d_monoD, d_monoQ, d_Bd, d_Bq, n_check_models, n_tiles, d_cnt,
                                                        d_und, d_dump);
        CHECK(hipDeviceSynchronize());
        std::vector<float> dump((size_t)n_pts * n_check_models * 5);
        CHECK(hipMemcpy(dump.data(), d_dump, dump.size() * 4, hipMemcpyDeviceToHost));
        long chain_mismatch = 0, wrong = 0, undecided = 0, n_in = 0, total = 0;
        double worst16 = 0;
        for (int i = 0; i < n_pts; ++i)
            for (int j = 0; j < n_check_models; ++j) {
                const float* o = &dump[((size_t)i * n_check_models + j) * 5];
                const int tile = i >> 5, r = i & 31;
                // (1) f32 MFMA against the k-ordered fmaf chain
                float acc = 0.f;
                for (int k = 0; k < 10; ++k)
                    acc = fmaf(monoD[((size_t)tile * 10 + k) * 32 + r], Bd[(size_t)j * 10 + k], acc);
                if (memcmp(&acc, &o[0], 4) != 0) ++chain_mismatch;
                // (2) f16 MFMA against the exact sum of the rounded operands
                for (int t = 0; t < 4; ++t) {
                    double ex = 0, sa = 0;
                    for (int k = 0; k < 16; ++k) {
                        const double p = (double)monoQ[((size_t)tile * 32 + r) * 16 + k] *
                                         (double)Bqc[((size_t)t * n_check_models + j) * 16 + k];
                        ex += p;
                        sa += fabs(p);
                    }
                    if (sa > 0) worst16 = fmax(worst16, fabs((double)o[1 + t] - ex) / sa);
                }
                // (3) verdicts
                const float n = o[0] * o[0];
                const bool in = n <= o[1] && n <= o[2], out = n >= o[3] || n >= o[4];
                const int ex = exact_inlier(&models[(size_t)j * 9], px[4 * i], px[4 * i + 1], px[4 * i + 2],
                                            px[4 * i + 3], (double)tf);
                if ((in && !ex) || (out && ex) || (in && out)) ++wrong;
                if (!in && !out) ++undecided;
                n_in += ex;
                ++total;
            }
        printf("numerics over %ld tests: f32 MFMA != fmaf chain: %ld; f16 MFMA worst |err|/sum|terms| = %.3g (2^%.1f);\n"
               "  verdicts contradicting double precision: %ld; undecided %.3f %%; inliers %.1f %%\n",
               total, chain_mismatch, worst16, log2(worst16 > 0 ? worst16 : 1e-300), wrong, 100.0 * undecided / total,
               100.0 * n_in / total);
        // subnormal operands: one column of coefficients 2^-20 against monomials 1
        {
            std::vector<_Float16> q(32 * 16, (_Float16)0.f), b(4 * 32 * 16, (_Float16)0.f);
            for (int r = 0; r < 32; ++r)
                for (int k = 0; k < 16; ++k) q[r * 16 + k] = (_Float16)1.0f;
            for (int j = 0; j < 32; ++j)
                for (int k = 0; k < 16; ++k) b[j * 16 + k] = (_Float16)ldexpf(1.f, -20);  // subnormal in f16
            CHECK(hipMemcpy(d_monoQ, q.data(), q.size() * 2, hipMemcpyHostToDevice));
            CHECK(hipMemcpy(d_Bq, b.data(), b.size() * 2, hipMemcpyHostToDevice));
            screen_kernel<true><<<1, 64>>>(d_monoD, d_monoQ, d_Bd, d_Bq, 32, 1, d_cnt, d_und, d_dump);
            CHECK(hipDeviceSynchronize());
            float o[5];
            CHECK(hipMemcpy(o, d_dump, 20, hipMemcpyDeviceToHost));
            printf("  sixteen products 1 * 2^-20 (f16 subnormal): MFMA gives %.9g, exact %.9g -> subnormal operands %s\n",
                   o[1], 16 * ldexp(1.0, -20), o[1] == (float)(16 * ldexp(1.0, -20)) ? "kept" : "FLUSHED/changed");
            CHECK(hipMemcpy(d_monoQ, monoQ.data(), monoQ.size() * 2, hipMemcpyHostToDevice));
        }
    }
    CHECK(hipMemcpy(d_Bq, Bq.data(), Bq.size() * 2, hipMemcpyHostToDevice));

    // ---- timing ----
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    const char* names[4] = {"matrix screen (f32 MFMA d)", "  its MFMAs only", "  its vector part only", "matrix screen, d by 3 f16 MFMAs"};
    for (int mode = 0; mode < 4; ++mode)
    for (int wps = 1; wps <= 4; wps *= 2) {
        const int waves = 1024 * wps;
        const int nm = waves * 32;  // screen: 32 models per wave
        float ms = 0;
        for (int rep = 0; rep < 2; ++rep) {
            hipMemset(d_cnt, 0, n_models * 4);
            hipMemset(d_und, 0, n_models * 4);
            hipEventRecord(a);
            if (mode == 0) screen_kernel<false, 0><<<waves, 64>>>(d_monoD, d_monoQ, d_Bd, d_Bq, n_models, n_tiles, d_cnt, d_und, nullptr);
            if (mode == 1) screen_kernel<false, 1><<<waves, 64>>>(d_monoD, d_monoQ, d_Bd, d_Bq, n_models, n_tiles, d_cnt, d_und, nullptr);
            if (mode == 2) screen_kernel<false, 2><<<waves, 64>>>(d_monoD, d_monoQ, d_Bd, d_Bq, n_models, n_tiles, d_cnt, d_und, nullptr);
            if (mode == 3) screen_kernel<false, 3><<<waves, 64>>>(d_monoD, d_monoQ, d_Bd, d_Bq, n_models, n_tiles, d_cnt, d_und, nullptr);
            hipEventRecord(b);
            hipEventSynchronize(b);
            hipEventElapsedTime(&ms, a, b);
        }
        const double tests = (double)nm * n_pts;
        printf("%-32s %d waves/SIMD: %8.1f us, %7.2f G tests/s\n", names[mode], wps, ms * 1e3, tests / (ms * 1e6));
    }
    for (int wps = 1; wps <= 8; wps *= 2) {
        const int waves = 1024 * wps;
        const int nm = waves * 64;  // vector filter: 64 models per wave
        if (nm > n_models) break;
        float ms = 0;
        for (int rep = 0; rep < 2; ++rep) {
            hipMemset(d_cnt, 0, n_models * 4);
            hipMemset(d_und, 0, n_models * 4);
            hipEventRecord(a);
            vector_kernel<<<waves, 64>>>(d_soa, d_Mf, d_vc, n_pts, d_cnt, d_und);
            hipEventRecord(b);
            hipEventSynchronize(b);
            hipEventElapsedTime(&ms, a, b);
        }
        const double tests = (double)nm * n_pts;
        printf("vector filter   %d waves/SIMD: %8.1f us, %7.2f G tests/s\n", wps, ms * 1e3, tests / (ms * 1e6));
    }
    return 0;
}
