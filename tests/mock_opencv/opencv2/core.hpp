// NOT OpenCV.  Declarations only, no behaviour: just enough of the names that
// include/sfm_gpu_cv.hpp and tools/opencv_crosscheck.cpp use for a compiler to
// parse those two files (-fsyntax-only) on a machine without OpenCV, so that a
// typo in them does not wait for the first box that has the real library.  It
// says nothing about OpenCV's semantics and nothing is ever linked or run
// against it (tests/test_abi.py).
#ifndef SFM_GPU_MOCK_OPENCV_CORE_HPP
#define SFM_GPU_MOCK_OPENCV_CORE_HPP
#include <cstddef>
#include <vector>

typedef unsigned char uchar;
#define CV_8U 0
#define CV_32F 5
#define CV_64F 6
#define CV_Assert(expr) do { if (!(expr)) cv::mock_fail(#expr); } while (0)
#define CV_Error(code, msg) cv::mock_fail(msg)

namespace cv {
[[noreturn]] void mock_fail(const char* what);
namespace Error { enum Code { StsError = -2, GpuNotSupported = -216 }; }
template <class T> struct Point_ {
    T x, y;
    Point_();
    Point_(T x_, T y_);
};
typedef Point_<double> Point2d;
template <class T> struct Point3_ {
    T x, y, z;
    Point3_();
    Point3_(T x_, T y_, T z_);
};
typedef Point3_<double> Point3d;
struct DMatch {
    int queryIdx, trainIdx, imgIdx;
    float distance;
};
struct MatStep {
    operator size_t() const;
};
class Mat {
public:
    Mat();
    Mat(int rows, int cols, int type);
    Mat(int rows, int cols, int type, void* data);
    int rows, cols;
    uchar* data;
    MatStep step;
    int type() const;
    bool empty() const;
    bool isContinuous() const;
    size_t total() const;
    Mat clone() const;
    Mat row(int i) const;
    void copyTo(Mat dst) const;
    void create(int rows, int cols, int type);
    template <class T> T& at(int i, int j);
    template <class T> const T& at(int i, int j) const;
    template <class T> T* ptr(int i = 0);
    template <class T> const T* ptr(int i = 0) const;
    static Mat zeros(int rows, int cols, int type);
};
Mat operator-(const Mat& a, const Mat& b);
Mat operator+(const Mat& a, const Mat& b);
double norm(const Mat& a);
template <class T> class Mat_ : public Mat {
public:
    Mat_(int rows, int cols);
};
template <class T> struct MatCommaInitializer_ {
    MatCommaInitializer_<T>& operator,(T v);
    operator Mat() const;
};
template <class T> MatCommaInitializer_<T> operator<<(const Mat_<T>& m, T v);
template <class T> MatCommaInitializer_<T> operator<<(const Mat_<T>& m, int v);
enum NormTypes { NORM_L2 = 4 };
}  // namespace cv
#endif
