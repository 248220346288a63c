// NOT OpenCV: see core.hpp beside this file.
#ifndef SFM_GPU_MOCK_OPENCV_FEATURES2D_HPP
#define SFM_GPU_MOCK_OPENCV_FEATURES2D_HPP
#include "core.hpp"
namespace cv {
class BFMatcher {
public:
    explicit BFMatcher(int normType = NORM_L2, bool crossCheck = false);
    void knnMatch(const Mat& query, const Mat& train, std::vector<std::vector<DMatch>>& matches, int k) const;
};
}  // namespace cv
#endif
