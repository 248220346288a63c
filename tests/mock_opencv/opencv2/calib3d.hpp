// NOT OpenCV: see core.hpp beside this file.
#ifndef SFM_GPU_MOCK_OPENCV_CALIB3D_HPP
#define SFM_GPU_MOCK_OPENCV_CALIB3D_HPP
#include "core.hpp"
namespace cv {
enum { LMEDS = 4, RANSAC = 8, FM_RANSAC = 8 };
enum SolvePnPMethod { SOLVEPNP_EPNP = 1 };
Mat findEssentialMat(const std::vector<Point2d>& p1, const std::vector<Point2d>& p2, const Mat& K, int method,
                     double prob, double threshold);
Mat findEssentialMat(const std::vector<Point2d>& p1, const std::vector<Point2d>& p2, const Mat& K, int method,
                     double prob, double threshold, std::vector<uchar>& mask);
Mat findFundamentalMat(const std::vector<Point2d>& p1, const std::vector<Point2d>& p2, int method, double threshold,
                       double confidence, std::vector<uchar>& mask);
int recoverPose(const Mat& E, const std::vector<Point2d>& p1, const std::vector<Point2d>& p2, const Mat& K, Mat& R,
                Mat& t);
bool solvePnPRansac(const std::vector<Point3d>& X, const std::vector<Point2d>& u, const Mat& K, const Mat& dist,
                    Mat& rvec, Mat& tvec, bool useExtrinsicGuess, int iterationsCount, float reprojectionError,
                    double confidence, std::vector<int>& inliers, int flags);
}  // namespace cv
#endif
