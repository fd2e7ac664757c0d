// tests/cpp/opencv_model/opencv2/core/core.hpp -- NOT OpenCV.  A declaration-level MODEL of the few cv:: types the facade's
// OpenCV branch touches (include/cvsteer/Mat.h: `typedef cv::Mat1f Mat1f; typedef cv::Point Point;`), written so that the
// branch goes through a compiler on an image that has no OpenCV: member NAMES AND TYPES follow OpenCV's core/mat.hpp /
// core/types.hpp as documented (cv::Mat::data is `uchar*`, cv::Mat::step is a `MatStep` that converts to size_t,
// `Mat::ptr(int)` returns `uchar*`, `Mat_<T>::operator()(int, int)`, `Mat_<T>::operator()(Point)`, `Mat_<T>::create(int, int)`,
// `Mat_<T>::clone()` returns Mat_<T>, copies are shallow and reference-counted, the (rows, cols, T*, step) constructor does
// not own).  Test scaffolding only (tests/test_facade_opencv_model.py puts this directory on the include path); nothing
// here is shipped or used by the product, and passing says "compiles and runs against these declarations", not "against
// OpenCV" -- tests/test_gpu_opencv.py and a build with the real headers remain the check the day a box has them.
#ifndef CVSTEER_TESTS_OPENCV_MODEL_CORE_HPP
#define CVSTEER_TESTS_OPENCV_MODEL_CORE_HPP
#include <cstddef>
#include <cstring>
#include <memory>

#define CVSTEER_TESTS_OPENCV_MODEL 1
#define CV_32F 5
#define CV_32FC1 5

namespace cv {

typedef unsigned char uchar;

template <typename T> struct Point_ {
    T x, y;
    Point_() : x(0), y(0) {}
    Point_(T x_, T y_) : x(x_), y(y_) {}
};
typedef Point_<int> Point2i;
typedef Point2i Point;

struct MatStep {
    size_t p[2];
    MatStep() { p[0] = p[1] = 0; }
    explicit MatStep(size_t s) { p[0] = s; p[1] = 0; }
    operator size_t() const { return p[0]; }
    MatStep& operator=(size_t s) { p[0] = s; return *this; }
    size_t& operator[](int i) { return p[i]; }
    const size_t& operator[](int i) const { return p[i]; }
};

class Mat {
public:
    enum { AUTO_STEP = 0 };
    int flags, dims, rows, cols;
    uchar* data;
    MatStep step;

    Mat() : flags(0), dims(0), rows(0), cols(0), data(0) {}
    Mat(int r, int c, int type) : flags(0), dims(0), rows(0), cols(0), data(0) { create(r, c, type); }
    Mat(int r, int c, int type, void* p, size_t s = AUTO_STEP) : flags(type), dims(2), rows(r), cols(c), data(static_cast<uchar*>(p))
    {
        esz_ = elem(type);
        step = s ? s : (size_t)c * esz_;
        step[1] = esz_;
    }
    int type() const { return flags; }
    size_t elemSize() const { return esz_; }
    bool empty() const { return data == 0 || rows == 0 || cols == 0; }
    size_t total() const { return (size_t)rows * cols; }
    bool isContinuous() const { return (size_t)step == (size_t)cols * esz_; }
    void create(int r, int c, int type)
    {
        if (r == rows && c == cols && type == flags && data) return;
        esz_ = elem(type);
        owner_.reset(new uchar[(size_t)r * c * esz_], std::default_delete<uchar[]>());
        flags = type;
        dims = 2;
        rows = r;
        cols = c;
        step = (size_t)c * esz_;
        step[1] = esz_;
        data = owner_.get();
    }
    void release()
    {
        owner_.reset();
        data = 0;
        rows = cols = 0;
    }
    uchar* ptr(int r = 0) { return data + (size_t)r * step; }
    const uchar* ptr(int r = 0) const { return data + (size_t)r * step; }
    template <typename T> T* ptr(int r = 0) { return reinterpret_cast<T*>(data + (size_t)r * step); }
    template <typename T> const T* ptr(int r = 0) const { return reinterpret_cast<const T*>(data + (size_t)r * step); }
    template <typename T> T& at(int r, int c) { return ptr<T>(r)[c]; }
    template <typename T> const T& at(int r, int c) const { return ptr<T>(r)[c]; }
    Mat clone() const
    {
        Mat m;
        if (empty()) return m;
        m.create(rows, cols, flags);
        for (int r = 0; r < rows; ++r) std::memcpy(m.ptr(r), ptr(r), (size_t)cols * esz_);
        return m;
    }

protected:
    static size_t elem(int type) { return type == CV_32F ? 4 : 1; }
    size_t esz_ = 0;
    std::shared_ptr<uchar> owner_;   // shallow copies share storage
};

template <typename T> struct DataType_;
template <> struct DataType_<float> { enum { type = CV_32F }; };

template <typename T> class Mat_ : public Mat {
public:
    Mat_() : Mat() { flags = DataType_<T>::type; }
    Mat_(int r, int c) : Mat(r, c, DataType_<T>::type) {}
    Mat_(int r, int c, T* p, size_t s = AUTO_STEP) : Mat(r, c, DataType_<T>::type, p, s) {}
    Mat_(const Mat& m) : Mat(m) {}   // (OpenCV converts when the type differs; the facade only ever passes f32)
    void create(int r, int c) { Mat::create(r, c, DataType_<T>::type); }
    Mat_ clone() const { return Mat_(Mat::clone()); }
    T* operator[](int r) { return Mat::ptr<T>(r); }
    const T* operator[](int r) const { return Mat::ptr<T>(r); }
    T& operator()(int r, int c) { return Mat::ptr<T>(r)[c]; }
    const T& operator()(int r, int c) const { return Mat::ptr<T>(r)[c]; }
    T& operator()(Point p) { return Mat::ptr<T>(p.y)[p.x]; }
    const T& operator()(Point p) const { return Mat::ptr<T>(p.y)[p.x]; }
};
typedef Mat_<float> Mat1f;

}  // namespace cv
#endif
