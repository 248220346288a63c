// opencv_crosscheck — runs the calls of the reference's matching path through
// real OpenCV and through libsfm_gpu on the same seeded inputs and reports every
// difference.  This is the check that pins the CPU oracle's restated OpenCV
// semantics (tie order, float distances, RNG stream, iteration counts, masks);
// it needs a machine with OpenCV >= 4.5 AND an MI355X, and is built only when
// CMake finds OpenCV (tools/CMakeLists.txt).  Nothing in the test suite depends
// on it.
//
// Calls compared (reference lines):
//   cv::BFMatcher(NORM_L2).knnMatch(q, t, matches, 2)      main.cpp:181-184 (exact brute force)
//   ratio test in double                                    main.cpp:189-197
//   cv::findEssentialMat(p1, p2, K, RANSAC, 0.99, 1.0, mask)      main.cpp:224
//   cv::findFundamentalMat(p1, p2, FM_RANSAC, 3.0, 0.99, mask)     (BASELINE's single-pair configuration)
//   cv::recoverPose(E, p1, p2, K, R, t)                            main.cpp:420
// and one experiment on OpenCV alone: which floating-point form findEssentialMat
// uses to normalise pixel coordinates (probe_normalisation below).
//   cv::solvePnPRansac(X, u, K, zeros, r, t, false, 100, 8, 0.99, inl, SOLVEPNP_EPNP)
//                                                        bundle_adjustment.h:154
#include <cmath>
#include <cstdio>
#include <cstring>
#include <random>
#include <vector>

#include <opencv2/calib3d.hpp>
#include <opencv2/core.hpp>
#include <opencv2/features2d.hpp>

#include "sfm_gpu_cv.hpp"

namespace {

// SIFT-like rows: 128 exponential magnitudes, L2-normalised, clipped at 0.2,
// renormalised, scaled by 512, rounded, saturated to 255 (CV_32F holding integers).
cv::Mat sift_like(int n, unsigned seed)
{
    std::mt19937_64 g(seed);
    std::exponential_distribution<double> e(1.0);
    cv::Mat m(n, 128, CV_32F);
    for (int i = 0; i < n; ++i) {
        double v[128], s = 0;
        for (double& x : v) { x = e(g); s += x * x; }
        s = std::sqrt(s);
        double s2 = 0;
        for (double& x : v) { x = std::min(x / s, 0.2); s2 += x * x; }
        s2 = std::sqrt(s2);
        for (int d = 0; d < 128; ++d) m.at<float>(i, d) = (float)std::min(255.0, std::nearbyint(v[d] / s2 * 512.0));
    }
    return m;
}

int diff_knn(const sfm_gpu_cv::Matcher& gpu, int nq, int nt, unsigned seed)
{
    cv::Mat q = sift_like(nq, seed), t = sift_like(nt, seed + 1);
    // planted near-duplicates and exact duplicates (tie order)
    for (int i = 0; i < std::min(nq, nt) / 3; ++i) q.row(i).copyTo(t.row((i * 7) % nt));
    if (nt > 8) t.row(3).copyTo(t.row(5));
    std::vector<std::vector<cv::DMatch>> a, b;
    cv::BFMatcher(cv::NORM_L2).knnMatch(q, t, a, 2);
    gpu.knnMatch(q, t, b, 2);
    int bad = a.size() != b.size();
    for (size_t i = 0; i < a.size() && i < b.size(); ++i) {
        if (a[i].size() != b[i].size()) { ++bad; continue; }
        for (size_t k = 0; k < a[i].size(); ++k)
            if (a[i][k].trainIdx != b[i][k].trainIdx || a[i][k].queryIdx != b[i][k].queryIdx ||
                std::memcmp(&a[i][k].distance, &b[i][k].distance, 4) != 0)
                ++bad;
    }
    std::printf("knnMatch %d x %d: %d differing records\n", nq, nt, bad);
    return bad;
}

void two_view(int n, double inlier_frac, unsigned seed, const cv::Mat& K, std::vector<cv::Point2d>& p1,
              std::vector<cv::Point2d>& p2)
{
    std::mt19937_64 g(seed);
    std::uniform_real_distribution<double> ux(0, 1600), uy(0, 1200), uz(425, 935);
    std::normal_distribution<double> nz(0, 0.5);
    const double fx = K.at<double>(0, 0), fy = K.at<double>(1, 1), cx = K.at<double>(0, 2), cy = K.at<double>(1, 2);
    const double a = 0.05, c = std::cos(a), s = std::sin(a);
    for (int i = 0; i < n; ++i) {
        const double u = ux(g), v = uy(g), z = uz(g);
        if (i < n * inlier_frac) {
            const double X = (u - cx) / fx * z, Y = (v - cy) / fy * z;
            const double X2 = c * X + s * z + 25.0, Z2 = -s * X + c * z + 2.0;
            p1.emplace_back((float)(u + nz(g)), (float)(v + nz(g)));
            p2.emplace_back((float)(X2 / Z2 * fx + cx + nz(g)), (float)(Y / Z2 * fy + cy + nz(g)));
        } else {
            p1.emplace_back((float)u, (float)v);
            p2.emplace_back((float)ux(g), (float)uy(g));
        }
    }
}

int diff_essential(const sfm_gpu_cv::Matcher& gpu, int n, double frac, unsigned seed)
{
    cv::Mat K = (cv::Mat_<double>(3, 3) << 2892.33, 0, 823.205, 0, 2883.18, 619.071, 0, 0, 1);  // main.cpp:103-105
    std::vector<cv::Point2d> p1, p2;
    two_view(n, frac, seed, K, p1, p2);
    std::vector<uchar> ma(n, 0), mb(n, 0);
    cv::Mat Ea = cv::findEssentialMat(p1, p2, K, cv::RANSAC, 0.99, 1.0, ma);
    cv::Mat Eb = gpu.findEssentialMat(p1, p2, K, 0.99, 1.0, &mb);
    int bad = 0;
    for (int i = 0; i < n; ++i) bad += (ma[i] != 0) != (mb[i] != 0);
    double de = -1;
    if (!Ea.empty() && !Eb.empty() && Ea.rows == 3 && Eb.rows == 3)
        de = std::min(cv::norm(Ea - Eb), cv::norm(Ea + Eb)) / cv::norm(Ea);
    std::printf("findEssentialMat n=%d inliers %.0f%%: %d mask bytes differ, |dE|/|E| = %.3g\n", n, frac * 100, bad, de);
    if (!Ea.empty() && !Eb.empty() && Ea.rows == 3) {
        cv::Mat Ra, ta, Rb, tb;
        const int ga = cv::recoverPose(Ea, p1, p2, K, Ra, ta);
        const int gb = gpu.recoverPose(Ea, p1, p2, K, Rb, tb);
        std::printf("recoverPose: votes %d vs %d, |dR| = %.3g, |dt| = %.3g\n", ga, gb, cv::norm(Ra - Rb), cv::norm(ta - tb));
        bad += ga != gb;
    }
    return bad;
}

int diff_fundamental(const sfm_gpu_cv::Matcher& gpu, int n, double frac, unsigned seed)
{
    cv::Mat K = (cv::Mat_<double>(3, 3) << 2892.33, 0, 823.205, 0, 2883.18, 619.071, 0, 0, 1);
    std::vector<cv::Point2d> p1, p2;
    two_view(n, frac, seed, K, p1, p2);
    std::vector<uchar> ma(n, 0), mb(n, 0);
    cv::Mat Fa = cv::findFundamentalMat(p1, p2, cv::FM_RANSAC, 3.0, 0.99, ma);
    cv::Mat Fb = gpu.findFundamentalMat(p1, p2, 3.0, 0.99, &mb);
    int bad = Fa.empty() != Fb.empty();
    for (int i = 0; i < n; ++i) bad += (ma[i] != 0) != (mb[i] != 0);
    double df = -1;
    if (!Fa.empty() && !Fb.empty() && Fa.rows == 3 && Fb.rows == 3) df = cv::norm(Fa - Fb) / cv::norm(Fa);
    std::printf("findFundamentalMat n=%d inliers %.0f%%: %d mask bytes differ, |dF|/|F| = %.3g\n", n, frac * 100, bad,
                df);
    return bad;
}

// Which arithmetic does cv::findEssentialMat use to normalise pixels?  The library
// (and its CPU oracle) compute x = (u - cx) / fx.  A matrix-expression evaluation
// of (points - pp) / focal may instead multiply by a reciprocal, x = u * (1/fx) +
// (-cx * (1/fx)).  The two differ in the last bit of some coordinates, which can
// move a borderline inlier.  OpenCV answers by itself: with an identity camera
// matrix its own normalisation is exact, so calling it on points normalised
// beforehand in either form and comparing the essential matrices of a five-point
// call (no RANSAC, every candidate returned) with those of the ordinary call on
// pixels shows which form the ordinary call applied.
void probe_normalisation()
{
    cv::Mat K = (cv::Mat_<double>(3, 3) << 2892.33, 0, 823.205, 0, 2883.18, 619.071, 0, 0, 1);
    cv::Mat I = (cv::Mat_<double>(3, 3) << 1, 0, 0, 0, 1, 0, 0, 0, 1);
    const double fx = 2892.33, fy = 2883.18, cx = 823.205, cy = 619.071;
    int same_div = 0, same_mul = 0, trials = 0;
    for (unsigned seed = 1; seed <= 50; ++seed) {
        std::vector<cv::Point2d> p1, p2, d1, d2, m1, m2;
        two_view(5, 1.0, 900 + seed, K, p1, p2);
        for (int i = 0; i < 5; ++i) {
            d1.emplace_back((p1[i].x - cx) / fx, (p1[i].y - cy) / fy);
            d2.emplace_back((p2[i].x - cx) / fx, (p2[i].y - cy) / fy);
            const double ix = 1.0 / fx, iy = 1.0 / fy;
            m1.emplace_back(p1[i].x * ix + (-cx * ix), p1[i].y * iy + (-cy * iy));
            m2.emplace_back(p2[i].x * ix + (-cx * ix), p2[i].y * iy + (-cy * iy));
        }
        const double thr = 1.0 / ((fx + fy) / 2);
        cv::Mat E = cv::findEssentialMat(p1, p2, K, cv::RANSAC, 0.99, 1.0);
        cv::Mat Ed = cv::findEssentialMat(d1, d2, I, cv::RANSAC, 0.99, thr);
        cv::Mat Em = cv::findEssentialMat(m1, m2, I, cv::RANSAC, 0.99, thr);
        if (E.empty()) continue;
        ++trials;
        same_div += E.rows == Ed.rows && cv::norm(E - Ed) == 0;
        same_mul += E.rows == Em.rows && cv::norm(E - Em) == 0;
    }
    std::printf("normalisation probe, %d five-point calls: identical to (u-cx)/fx in %d, to u*(1/fx)+(-cx/fx) in %d\n"
                "  (the library and its oracle use the first form)\n", trials, same_div, same_mul);
}

int diff_pnp(const sfm_gpu_cv::Matcher& gpu, int n, double out_frac, unsigned seed)
{
    cv::Mat K = (cv::Mat_<double>(3, 3) << 2892.33, 0, 823.205, 0, 2883.18, 619.071, 0, 0, 1);
    std::mt19937_64 g(seed);
    std::uniform_real_distribution<double> ux(-150, 150), up(0, 1500);
    std::normal_distribution<double> nz(0, 0.5);
    const double a = 0.3, c = std::cos(a), s = std::sin(a);
    std::vector<cv::Point3d> X;
    std::vector<cv::Point2d> u;
    for (int i = 0; i < n; ++i) {
        const double x = ux(g), y = ux(g), z = ux(g);
        X.emplace_back(x, y, z);
        const double xc = c * x + s * z + 10, yc = y - 20, zc = -s * x + c * z + 640;
        if (i < n * out_frac) u.emplace_back(up(g), up(g));
        else u.emplace_back(xc / zc * 2892.33 + 823.205 + nz(g), yc / zc * 2883.18 + 619.071 + nz(g));
    }
    cv::Mat ra, ta, rb, tb;
    std::vector<int> ia, ib;
    const bool oka = cv::solvePnPRansac(X, u, K, cv::Mat::zeros(1, 5, CV_64F), ra, ta, false, 100, 8.0f, 0.99, ia,
                                        cv::SOLVEPNP_EPNP);
    const bool okb = gpu.solvePnPRansac(X, u, K, rb, tb, 100, 8.0f, 0.99, &ib);
    std::printf("solvePnPRansac n=%d outliers %.0f%%: ok %d/%d, inliers %zu/%zu%s, |dr| = %.3g, |dt| = %.3g\n", n,
                out_frac * 100, (int)oka, (int)okb, ia.size(), ib.size(), ia == ib ? " (identical)" : " (DIFFER)",
                oka && okb ? cv::norm(ra - rb) : -1.0, oka && okb ? cv::norm(ta - tb) : -1.0);
    return (oka != okb) + (ia != ib);
}

}  // namespace

int main()
{
    sfm_gpu_cv::Matcher gpu(0);
    int bad = 0;
    for (auto [nq, nt] : {std::pair<int, int>{1, 1}, {5, 2}, {257, 1000}, {4096, 4096}, {8192, 8192}})
        bad += diff_knn(gpu, nq, nt, 100 + nq);
    for (auto [n, f] : {std::pair<int, double>{5, 1.0}, {64, 0.9}, {3000, 0.9}, {3000, 0.5}, {12000, 0.3}})
        bad += diff_essential(gpu, n, f, 200 + n);
    for (auto [n, f] : {std::pair<int, double>{7, 1.0}, {64, 0.9}, {3000, 0.9}, {3000, 0.5}})
        bad += diff_fundamental(gpu, n, f, 400 + n);
    for (auto [n, f] : {std::pair<int, double>{5, 0.0}, {40, 0.2}, {3000, 0.3}})
        bad += diff_pnp(gpu, n, f, 300 + n);
    probe_normalisation();
    std::printf("%s: %d differences\n", bad ? "DIFFERENT" : "identical", bad);
    return bad ? 1 : 0;
}
