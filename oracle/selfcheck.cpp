/*
 * CPU ORACLE — TEST INFRASTRUCTURE ONLY.
 *
 * Stand-alone run of every oracle entry point on small known cases, meant to be
 * built with -fsanitize=address,undefined (`make asan`): out-of-bounds accesses,
 * leaks and undefined arithmetic in the checker show up here rather than as a
 * silently wrong "expected" value in a parity test.  Exit code 0 = all cases
 * behaved; any sanitizer report aborts the program.
 *
 * The expected values are the known answers that need no OpenCV: hand distances,
 * the cv::RNG recurrence from seed 2^64-1, the iteration-count formula
 * (what /root/reference/main.cpp:184 and :224 reach inside OpenCV 4.5.5).
 */
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "sfm_oracle.h"

static int failures = 0;
#define CHECK(cond)                                                         \
    do {                                                                    \
        if (!(cond)) {                                                      \
            std::fprintf(stderr, "selfcheck: %s:%d: %s\n", __FILE__, __LINE__, #cond); \
            ++failures;                                                     \
        }                                                                   \
    } while (0)

// splitmix64: inputs only, no claim about any library
static uint64_t sm_state = 0x1234abcdULL;
static uint64_t sm_next()
{
    uint64_t z = (sm_state += 0x9E3779B97F4A7C15ULL);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}
static double sm_unit() { return (double)(sm_next() >> 11) / 9007199254740992.0; }

static void rot_small(double ax, double ay, double az, double* R)
{
    const double th = std::sqrt(ax * ax + ay * ay + az * az);
    const double c = std::cos(th), s = std::sin(th), x = ax / th, y = ay / th, z = az / th;
    const double M[9] = {c + x * x * (1 - c),     x * y * (1 - c) - z * s, x * z * (1 - c) + y * s,
                         y * x * (1 - c) + z * s, c + y * y * (1 - c),     y * z * (1 - c) - x * s,
                         z * x * (1 - c) - y * s, z * y * (1 - c) + x * s, c + z * z * (1 - c)};
    std::memcpy(R, M, sizeof(M));
}

int main()
{
    // ---- kNN: hand distances, tie order, short train set, both code paths, strided rows
    {
        const int dim = 128, nq = 9, nt = 7, stride = dim + 3;
        std::vector<float> Q((size_t)nq * stride, 0.f), T((size_t)nt * stride, 0.f);
        for (int i = 0; i < nq; i++)
            for (int d = 0; d < dim; d++) Q[(size_t)i * stride + d] = (float)(sm_next() % 256);
        for (int j = 0; j < nt; j++)
            for (int d = 0; d < dim; d++) T[(size_t)j * stride + d] = (float)(sm_next() % 256);
        // train rows 2 and 5 equal query row 0 but for one coordinate off by 1 and 2
        for (int d = 0; d < dim; d++) T[2 * stride + d] = T[5 * stride + d] = Q[d];
        T[2 * stride + 7] = Q[7] < 254 ? Q[7] + 2 : Q[7] - 2;
        T[5 * stride + 9] = Q[9] < 255 ? Q[9] + 1 : Q[9] - 1;
        // rows 3 and 4 duplicate each other: ties keep ascending train index
        for (int d = 0; d < dim; d++) T[4 * stride + d] = T[3 * stride + d];
        for (int d = 0; d < dim; d++) Q[1 * stride + d] = T[3 * stride + d];
        std::vector<oracle_dmatch> a((size_t)nq * 2), b((size_t)nq * 2);
        std::vector<int32_t> ca(nq), cb(nq);
        CHECK(oracle_knn_match(Q.data(), nq, stride * 4, T.data(), nt, stride * 4, dim, 2, a.data(), ca.data(), 3) == 0);
        CHECK(oracle_knn_match_u8path(Q.data(), nq, stride * 4, T.data(), nt, stride * 4, dim, 2, b.data(), cb.data(),
                                      2) == 0);
        CHECK(std::memcmp(a.data(), b.data(), a.size() * sizeof(oracle_dmatch)) == 0);
        CHECK(std::memcmp(ca.data(), cb.data(), nq * 4) == 0);
        CHECK(a[0].trainIdx == 5 && a[0].distance == 1.0f && a[1].trainIdx == 2 && a[1].distance == 2.0f);
        CHECK(a[2].trainIdx == 3 && a[3].trainIdx == 4 && a[2].distance == 0.0f && a[3].distance == 0.0f);
        // one train row: rows of length 1; no train rows: counts 0
        CHECK(oracle_knn_match_u8path(Q.data(), nq, stride * 4, T.data(), 1, stride * 4, dim, 2, b.data(), cb.data(),
                                      1) == 0);
        CHECK(cb[0] == 1 && b[0].trainIdx == 0 && b[1].trainIdx == -1);
        CHECK(oracle_knn_match(Q.data(), nq, stride * 4, T.data(), 0, stride * 4, dim, 2, a.data(), ca.data(), 1) == 0);
        CHECK(ca[0] == 0 && a[0].trainIdx == -1);
        // a value that is not a byte falls through to the float path, same result
        const float q5 = Q[5];
        Q[5] = 100.5f;
        CHECK(oracle_knn_match(Q.data(), nq, stride * 4, T.data(), nt, stride * 4, dim, 2, a.data(), ca.data(), 1) == 0);
        CHECK(oracle_knn_match_u8path(Q.data(), nq, stride * 4, T.data(), nt, stride * 4, dim, 2, b.data(), cb.data(),
                                      4) == 0);
        CHECK(std::memcmp(a.data(), b.data(), a.size() * sizeof(oracle_dmatch)) == 0);
        Q[5] = q5;
        // ratio test in double
        CHECK(oracle_knn_match(Q.data(), nq, stride * 4, T.data(), nt, stride * 4, dim, 2, a.data(), ca.data(), 1) == 0);
        std::vector<int32_t> keep(nq);
        const int nk = oracle_ratio_test(a.data(), ca.data(), nq, 2, 0.7, keep.data());
        CHECK(nk >= 1 && keep[0] == 0);   // 1 < 0.7 * 2
    }
    // ---- cv::RNG from seed 2^64-1
    {
        uint32_t r[8];
        oracle_rng_next(0xFFFFFFFFFFFFFFFFULL, 8, r);
        const uint32_t want[8] = {0x07c09cf5u, 0xbac3439cu, 0x99ae7b8cu, 0x37275f45u,
                                  0x40637cc7u, 0xaf98a118u, 0x059ae535u, 0x8902c9a0u};
        for (int i = 0; i < 8; i++) CHECK(r[i] == want[i]);
        int32_t u[8];
        oracle_rng_uniform(0xFFFFFFFFFFFFFFFFULL, 0, 1000, 8, u);
        const int32_t wu[8] = {605, 4, 940, 173, 831, 512, 301, 280};
        for (int i = 0; i < 8; i++) CHECK(u[i] == wu[i]);
        CHECK(oracle_ransac_update_num_iters(0.99, 1 - 0.91, 5, 1000) == 5);
        std::vector<int32_t> tab(40 * 5);
        CHECK(oracle_sample_table(37, 5, 40, tab.data()) == 0);
        for (int i = 0; i < 40 * 5; i++) CHECK(tab[i] >= 0 && tab[i] < 37);
        CHECK(oracle_sample_table(4, 5, 1, tab.data()) == -1);
    }
    // ---- two-view geometry: RANSAC in both models and modes, pose, PnP, joins
    {
        const double K[9] = {2892.33, 0, 823.205, 0, 2883.18, 619.071, 0, 0, 1};
        const int n = 300;
        double R[9];
        rot_small(0.02, -0.03, 0.01, R);
        const double t[3] = {20.0, -3.0, 2.0};
        std::vector<double> p1((size_t)n * 2), p2((size_t)n * 2), X((size_t)n * 3);
        for (int i = 0; i < n; i++) {
            const double u = sm_unit() * 1600, v = sm_unit() * 1200, z = 425 + sm_unit() * 510;
            const double x = (u - K[2]) / K[0] * z, y = (v - K[5]) / K[4] * z;
            X[3 * i] = x; X[3 * i + 1] = y; X[3 * i + 2] = z;
            const double a = R[0] * x + R[1] * y + R[2] * z + t[0], b = R[3] * x + R[4] * y + R[5] * z + t[1],
                         c = R[6] * x + R[7] * y + R[8] * z + t[2];
            p1[2 * i] = (double)(float)u;
            p1[2 * i + 1] = (double)(float)v;
            p2[2 * i] = (double)(float)(K[0] * a / c + K[2] + (sm_unit() - 0.5));
            p2[2 * i + 1] = (double)(float)(K[4] * b / c + K[5] + (sm_unit() - 0.5));
            if (i % 5 == 0) { p2[2 * i] = sm_unit() * 1600; p2[2 * i + 1] = sm_unit() * 1200; }
        }
        for (int model = 0; model < 2; model++)
            for (int fixed = 0; fixed <= 64; fixed += 64) {
                std::vector<uint8_t> mask(n, 0), mask_mt(n, 0);
                double M[9], Mmt[9];
                int32_t ninl = 0, iters = 0, ninl_mt = 0;
                std::vector<int32_t> hc((size_t)1000 * 10);
                const double thr = model ? 3.0 : 1.0;
                const int ok = oracle_ransac(model, p1.data(), p2.data(), n, K, 0.99, thr, 1000, fixed, nullptr,
                                             mask.data(), M, &ninl, &iters, hc.data());
                CHECK(ok == 1 && ninl > n / 2);
                if (fixed) {
                    CHECK(oracle_ransac_fixed_mt(model, p1.data(), p2.data(), n, K, thr, fixed, 3, mask_mt.data(), Mmt,
                                                 &ninl_mt) == 1);
                    CHECK(ninl_mt == ninl && std::memcmp(mask.data(), mask_mt.data(), n) == 0 &&
                          std::memcmp(M, Mmt, sizeof(M)) == 0);
                }
                if (model == 0) {
                    double Rr[9], tr[3];
                    int32_t votes[4];
                    std::vector<uint8_t> pm(mask);
                    const int good = oracle_recover_pose(M, p1.data(), p2.data(), n, K, 50.0, Rr, tr, pm.data(), votes);
                    CHECK(good > n / 2);
                    double U[9], S[3], V[9], R1[9], R2[9], td[3];
                    CHECK(oracle_svd3(M, U, S, V) == 1);
                    CHECK(oracle_decompose_essential(M, R1, R2, td) == 1);
                }
            }
        // too few points, exactly m points
        {
            uint8_t mask[7] = {9, 9, 9, 9, 9, 9, 9};
            double M[9];
            int32_t ninl = -1, iters = -1;
            CHECK(oracle_ransac(0, p1.data() + 2, p2.data() + 2, 4, K, 0.99, 1.0, 1000, 0, nullptr, mask, M, &ninl,
                                &iters, nullptr) == 0);
            CHECK(mask[0] == 9);
            const int ok5 = oracle_ransac(0, p1.data() + 2, p2.data() + 2, 5, K, 0.99, 1.0, 1000, 0, nullptr, mask, M,
                                          &ninl, &iters, nullptr);
            CHECK(ok5 == 0 || (mask[0] == 1 && mask[4] == 1));
            double stack[10 * 9];
            (void)oracle_ransac_direct_models(stack, 10);
        }
        // resection on the planted pose
        {
            double rvec[3], tvec[3], Rp[9];
            std::vector<uint8_t> mask(n, 0);
            int32_t ninl = 0, iters = 0;
            CHECK(oracle_pnp_ransac(X.data(), p2.data(), n, K, 100, 8.0, 0.99, rvec, tvec, Rp, mask.data(), &ninl,
                                    &iters) == 1);
            CHECK(ninl > n / 2 && std::fabs(tvec[0] - t[0]) < 2.0);
            double Re[9], te[3];
            CHECK(oracle_epnp(X.data() + 3, p1.data() + 2, 8, K, Re, te) >= 0);
            CHECK(oracle_rotation_vector(Rp, rvec) >= 0);
        }
        // solvers and error functions on a minimal sample
        {
            double x1[14], x2[14], models[90];
            double thr = oracle_normalize_points(p1.data() + 2, 7, K, 1.0, x1);
            (void)oracle_normalize_points(p2.data() + 2, 7, K, 1.0, x2);
            CHECK(thr > 0);
            const int nm = oracle_five_point(x1, x2, models);
            CHECK(nm >= 0 && nm <= 10);
            float err[7];
            if (nm > 0) oracle_sampson_errors(models, x1, x2, 7, err);
            float f1[14], f2[14];
            for (int i = 0; i < 14; i++) { f1[i] = (float)p1[2 + i]; f2[i] = (float)p2[2 + i]; }
            const int n7 = oracle_seven_point(f1, f2, models);
            CHECK(n7 >= 0 && n7 <= 3);
            if (n7 > 0) oracle_symepi_errors(models, f1, f2, 7, err);
            const double coef[4] = {-6, 11, -6, 1};   // (z-1)(z-2)(z-3)
            double roots[10];
            CHECK(oracle_real_roots(coef, 3, roots) == 3 && std::fabs(roots[0] - 1) < 1e-9 &&
                  std::fabs(roots[2] - 3) < 1e-9);
            std::vector<float> g1((size_t)n * 2), g2((size_t)n * 2);
            for (int i = 0; i < 2 * n; i++) { g1[i] = (float)p1[i]; g2[i] = (float)p2[i]; }
            std::vector<int32_t> tab(50 * 7);
            CHECK(oracle_sample_table_f(n, g1.data(), g2.data(), 50, tab.data()) == 50);
            CHECK(oracle_sample_table_fixed(n, 7, 50, g1.data(), g2.data(), tab.data()) >= 0);
            CHECK(oracle_sample_table_fixed(n, 5, 50, nullptr, nullptr, tab.data()) >= 0);
        }
        // three-view join, with a capacity smaller than the result
        {
            oracle_dmatch ab[4] = {{0, 5, 0, 1.f}, {1, 7, 0, 1.f}, {2, 5, 0, 1.f}, {3, 9, 0, 1.f}};
            oracle_dmatch bc[3] = {{5, 11, 0, 1.f}, {5, 12, 0, 1.f}, {9, 13, 0, 1.f}};
            int32_t out[5 * 3];
            CHECK(oracle_track_join(ab, 4, bc, 3, out, 5) == 5);
            CHECK(out[0] == 0 && out[1] == 5 && out[2] == 11 && out[3 * 4] == 3 && out[3 * 4 + 2] == 13);
            int32_t small[2 * 3];
            CHECK(oracle_track_join(ab, 4, bc, 3, small, 2) == 5);
        }
    }
    if (failures) {
        std::fprintf(stderr, "selfcheck: %d check(s) failed\n", failures);
        return 1;
    }
    std::printf("oracle selfcheck ok\n");
    return 0;
}
