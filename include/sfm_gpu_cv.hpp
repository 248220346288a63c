// sfm_gpu_cv.hpp — header-only C++ adapter between OpenCV's types and the C ABI
// of libsfm_gpu (include/sfm_gpu.h).  Compiled only by hosts that have OpenCV:
// the library itself never includes it.
//
// Each method stands for one call of the reference (Suxilan/SFM), same
// arguments, same outputs:
//   knnMatch          matcher->knnMatch(descL, descR, matches, 2)        main.cpp:181-184
//   findEssentialMat  findEssentialMat(p1, p2, K, FM_RANSAC, prob, thr, mask)
//                                                     main.cpp:224, :418; bundle_adjustment.h:79
//   findFundamentalMat  findFundamentalMat(p1, p2, FM_RANSAC, thr, confidence, mask): the call BASELINE's
//                     single-pair configuration names; the reference itself only calls findEssentialMat
//   recoverPose       recoverPose(E, p1, p2, K, R, T)       main.cpp:420; bundle_adjustment.h:80
//   trackJoin         the nested loop over two match lists              main.cpp:442-456
//   trackJoinSequence that loop for every image of the sequence         main.cpp:432-456
//   solvePnPRansac    solvePnPRansac(pts3d, pts, K, dist, r, t, false, SOLVEPNP_EPNP)
//                                                              bundle_adjustment.h:139-155
// Errors of the C ABI surface as cv::Exception, like the calls they replace.
#ifndef SFM_GPU_CV_HPP
#define SFM_GPU_CV_HPP

#include <algorithm>
#include <cstring>
#include <vector>

#include <opencv2/core.hpp>

#include "sfm_gpu.h"

namespace sfm_gpu_cv {

class Matcher {
public:
    explicit Matcher(int device = 0)
    {
        if (sfm_gpu_create(device, nullptr, &ctx_) != SFM_GPU_OK)
            CV_Error(cv::Error::GpuNotSupported, "sfm_gpu_create failed (no MI355X / HIP runtime?)");
    }
    ~Matcher() { sfm_gpu_destroy(ctx_); }
    Matcher(const Matcher&) = delete;
    Matcher& operator=(const Matcher&) = delete;
    sfm_gpu_ctx* context() const { return ctx_; }

    // main.cpp:181-184.  Rows of `matches` are as long as the neighbours that exist
    // (min(k, train.rows)), like cv::DescriptorMatcher::knnMatch.
    void knnMatch(const cv::Mat& query, const cv::Mat& train, std::vector<std::vector<cv::DMatch>>& matches,
                  int k = 2) const
    {
        static_assert(sizeof(cv::DMatch) == sizeof(sfm_gpu_dmatch), "cv::DMatch layout");
        CV_Assert(query.type() == train.type() && (query.type() == CV_32F || query.type() == CV_8U));
        CV_Assert(query.cols == 128 && (train.empty() || train.cols == 128));
        matches.clear();
        if (query.empty() || train.empty()) return;  // OpenCV: empty result
        std::vector<sfm_gpu_dmatch> flat((size_t)query.rows * k);
        std::vector<int32_t> counts(query.rows);
        check(sfm_gpu_knn_match(ctx_, query.data, query.rows, query.step, train.data, train.rows, train.step, 128,
                                query.type() == CV_8U ? SFM_GPU_DESC_U8 : SFM_GPU_DESC_F32, k, flat.data(),
                                counts.data()));
        matches.resize(query.rows);
        for (int i = 0; i < query.rows; ++i) {
            matches[i].resize(counts[i]);
            if (counts[i] > 0)
                std::memcpy((void*)matches[i].data(), &flat[(size_t)i * k], counts[i] * sizeof(cv::DMatch));
        }
    }

    // main.cpp:224 (mask wanted), main.cpp:418 and bundle_adjustment.h:79 (mask == nullptr).
    // With exactly five points the result stacks every candidate (3k x 3), as OpenCV's does.
    cv::Mat findEssentialMat(const std::vector<cv::Point2d>& p1, const std::vector<cv::Point2d>& p2,
                             const cv::Mat& K, double prob = 0.999, double threshold = 1.0,
                             std::vector<uchar>* mask = nullptr, int maxIters = 1000) const
    {
        CV_Assert(K.type() == CV_64F && K.isContinuous() && K.total() == 9 && p1.size() == p2.size());
        double E[9];
        int32_t ok = 0;
        std::vector<uchar> m(p1.size(), 0);
        const int n = (int)p1.size();
        check(sfm_gpu_ransac(ctx_, n ? &p1[0].x : nullptr, n ? &p2[0].x : nullptr, n, K.ptr<double>(),
                             SFM_GPU_MODEL_ESSENTIAL, prob, threshold, maxIters, SFM_GPU_RANSAC_ADAPTIVE, nullptr, 0,
                             m.data(), E, nullptr, nullptr, &ok));
        if (!ok) return cv::Mat();  // the caller's mask stays as it was, like OpenCV's
        if (mask) *mask = m;
        if (n == 5) {
            double all[90];
            int32_t k = 0;
            check(sfm_gpu_ransac_candidates(ctx_, all, 10, &k));
            return cv::Mat(3 * k, 3, CV_64F, all).clone();
        }
        return cv::Mat(3, 3, CV_64F, E).clone();
    }

    // cv::findFundamentalMat(points1, points2, FM_RANSAC, ransacReprojThreshold, confidence, mask):
    // seven-point samples, symmetric epipolar distance, F33 = 1.  With exactly seven points the
    // result stacks every candidate (3k x 3); fewer than seven give an empty matrix; 8..14 points
    // make OpenCV switch to LMedS, which this path does not have (cv::Exception).
    cv::Mat findFundamentalMat(const std::vector<cv::Point2d>& p1, const std::vector<cv::Point2d>& p2,
                               double ransacReprojThreshold = 3.0, double confidence = 0.99,
                               std::vector<uchar>* mask = nullptr, int maxIters = 1000) const
    {
        CV_Assert(p1.size() == p2.size());
        double F[9];
        int32_t ok = 0;
        std::vector<uchar> m(p1.size(), 0);
        const int n = (int)p1.size();
        check(sfm_gpu_ransac(ctx_, n ? &p1[0].x : nullptr, n ? &p2[0].x : nullptr, n, nullptr,
                             SFM_GPU_MODEL_FUNDAMENTAL, confidence, ransacReprojThreshold, maxIters,
                             SFM_GPU_RANSAC_ADAPTIVE, nullptr, 0, m.data(), F, nullptr, nullptr, &ok));
        if (!ok) return cv::Mat();
        if (mask) *mask = m;
        if (n == 7) {
            double all[27];
            int32_t k = 0;
            check(sfm_gpu_ransac_candidates(ctx_, all, 3, &k));
            return cv::Mat(3 * k, 3, CV_64F, all).clone();
        }
        return cv::Mat(3, 3, CV_64F, F).clone();
    }

    // main.cpp:420, bundle_adjustment.h:80
    int recoverPose(const cv::Mat& E, const std::vector<cv::Point2d>& p1, const std::vector<cv::Point2d>& p2,
                    const cv::Mat& K, cv::Mat& R, cv::Mat& t, double distanceThresh = 50.0,
                    std::vector<uchar>* mask = nullptr) const
    {
        CV_Assert(E.type() == CV_64F && E.isContinuous() && E.total() == 9);
        CV_Assert(K.type() == CV_64F && K.isContinuous() && K.total() == 9 && p1.size() == p2.size());
        R.create(3, 3, CV_64F);
        t.create(3, 1, CV_64F);
        int32_t good = 0;
        if (mask && mask->size() != p1.size()) mask->assign(p1.size(), 1);
        const int n = (int)p1.size();
        check(sfm_gpu_recover_pose(ctx_, E.ptr<double>(), n ? &p1[0].x : nullptr, n ? &p2[0].x : nullptr, n,
                                   K.ptr<double>(), distanceThresh, R.ptr<double>(), t.ptr<double>(),
                                   mask ? mask->data() : nullptr, &good));
        return good;
    }

    // main.cpp:442-456: kps1/kps2/kps3 are keypoints[i][t.a], keypoints[i+1][t.b], keypoints[j][t.c]
    std::vector<sfm_gpu_track3> trackJoin(const std::vector<cv::DMatch>& ab, const std::vector<cv::DMatch>& bc) const
    {
        std::vector<sfm_gpu_track3> t(ab.size() + 1);
        int32_t n = 0;
        int rc = sfm_gpu_track_join(ctx_, (const sfm_gpu_dmatch*)ab.data(), (int)ab.size(),
                                    (const sfm_gpu_dmatch*)bc.data(), (int)bc.size(), t.data(), (int)t.size(), &n);
        if (rc == SFM_GPU_ERR_CAPACITY) {  // repeated query rows in bc: more triples than matches
            t.resize(n);
            rc = sfm_gpu_track_join(ctx_, (const sfm_gpu_dmatch*)ab.data(), (int)ab.size(),
                                    (const sfm_gpu_dmatch*)bc.data(), (int)bc.size(), t.data(), n, &n);
        }
        check(rc);
        t.resize(n);
        return t;
    }

    // The loop of main.cpp:432-456 for a whole sequence: matches[l] against matches[l+1] for every l,
    // in one call.  tracks[l] are the triples of images (l, l+1, j).
    std::vector<std::vector<sfm_gpu_track3>> trackJoinSequence(const std::vector<std::vector<cv::DMatch>>& matches) const
    {
        const int n_lists = (int)matches.size();
        std::vector<std::vector<sfm_gpu_track3>> tracks(n_lists > 1 ? n_lists - 1 : 0);
        if (n_lists < 2) return tracks;
        size_t cap = 1, room = 1;
        for (const auto& m : matches) cap = std::max(cap, m.size());
        for (int l = 0; l + 1 < n_lists; ++l) room += matches[l].size();
        std::vector<sfm_gpu_dmatch> packed(cap * n_lists);
        std::vector<int32_t> sizes(n_lists), offs(n_lists, 0);
        for (int l = 0; l < n_lists; ++l) {
            sizes[l] = (int32_t)matches[l].size();
            if (sizes[l]) std::memcpy(&packed[cap * l], matches[l].data(), sizeof(sfm_gpu_dmatch) * matches[l].size());
        }
        std::vector<sfm_gpu_track3> flat(room);
        int rc = sfm_gpu_track_join_sequence(ctx_, packed.data(), sizes.data(), (int)cap, n_lists, flat.data(),
                                             (int)flat.size(), offs.data());
        if (rc == SFM_GPU_ERR_CAPACITY) {  // repeated query rows: more triples than matches
            flat.resize(offs[n_lists - 1]);
            rc = sfm_gpu_track_join_sequence(ctx_, packed.data(), sizes.data(), (int)cap, n_lists, flat.data(),
                                             (int)flat.size(), offs.data());
        }
        check(rc);
        for (int l = 0; l + 1 < n_lists; ++l) tracks[l].assign(flat.begin() + offs[l], flat.begin() + offs[l + 1]);
        return tracks;
    }

    // bundle_adjustment.h:154 (distortion coefficients are all zero there and are not taken)
    bool solvePnPRansac(const std::vector<cv::Point3d>& objectPoints, const std::vector<cv::Point2d>& imagePoints,
                        const cv::Mat& K, cv::Mat& rvec, cv::Mat& tvec, int iterationsCount = 100,
                        float reprojectionError = 8.0f, double confidence = 0.99,
                        std::vector<int>* inliers = nullptr) const
    {
        CV_Assert(K.type() == CV_64F && K.isContinuous() && K.total() == 9);
        CV_Assert(objectPoints.size() == imagePoints.size() && objectPoints.size() >= 4);
        const int n = (int)objectPoints.size();
        rvec.create(3, 1, CV_64F);
        tvec.create(3, 1, CV_64F);
        std::vector<uchar> mask(n, 0);
        int32_t ok = 0;
        check(sfm_gpu_pnp_ransac(ctx_, &objectPoints[0].x, &imagePoints[0].x, n, K.ptr<double>(), iterationsCount,
                                 reprojectionError, confidence, rvec.ptr<double>(), tvec.ptr<double>(), nullptr,
                                 mask.data(), nullptr, nullptr, &ok));
        if (inliers) {
            inliers->clear();
            for (int i = 0; i < n && ok; ++i)
                if (mask[i]) inliers->push_back(i);
        }
        return ok != 0;
    }

private:
    void check(int rc) const
    {
        if (rc != SFM_GPU_OK) CV_Error(cv::Error::StsError, sfm_gpu_last_error_string(ctx_));
    }
    sfm_gpu_ctx* ctx_ = nullptr;
};

}  // namespace sfm_gpu_cv

#endif
