// cvsteer/Mat.h -- the matrix type of the facade.
//
// The reference's public surface is written in terms of cv::Mat1f / cv::Point
// (cvsteer/SteerableFilters.h:37,44-49).  When OpenCV headers are available (and
// CVSTEER_NO_OPENCV is not defined) fa::Mat1f IS cv::Mat1f and existing callers compile
// unchanged.  Without OpenCV (this build image has none) a minimal reference-counted f32
// matrix with the members the reference's callers use (rows, cols, step, data, create, empty,
// total, clone, operator()(row, col), operator()(Point)) stands in, so the same facade, tests
// and examples build and run.
#ifndef CVSTEER_AMD_MAT_H
#define CVSTEER_AMD_MAT_H

#if !defined(CVSTEER_NO_OPENCV) && defined(__has_include)
#if __has_include(<opencv2/core/core.hpp>)
#define CVSTEER_HAVE_OPENCV 1
#endif
#endif

#ifdef CVSTEER_HAVE_OPENCV
#include <opencv2/core/core.hpp>
namespace fa {
typedef cv::Mat1f Mat1f;
typedef cv::Point Point;
}
#else
#include <cstddef>
#include <cstring>
#include <memory>
namespace fa {

struct Point {
    int x, y;
    Point() : x(0), y(0) {}
    Point(int x_, int y_) : x(x_), y(y_) {}
};

class Mat1f {
public:
    int rows, cols;
    size_t step;  // bytes between rows
    float* data;

    Mat1f() : rows(0), cols(0), step(0), data(0) {}
    Mat1f(int r, int c) : rows(0), cols(0), step(0), data(0) { create(r, c); }
    // view of caller-owned memory (like cv::Mat(rows, cols, type, data, step)): not freed
    Mat1f(int r, int c, float* p, size_t step_bytes = 0)
        : rows(r), cols(c), step(step_bytes ? step_bytes : (size_t)c * sizeof(float)), data(p) {}

    void create(int r, int c)
    {
        if (r == rows && c == cols && data && owner_) return;
        owner_.reset(new float[(size_t)r * c], std::default_delete<float[]>());
        rows = r;
        cols = c;
        step = (size_t)c * sizeof(float);
        data = owner_.get();
    }
    bool empty() const { return data == 0 || rows == 0 || cols == 0; }
    size_t total() const { return (size_t)rows * cols; }
    float* ptr(int r) { return reinterpret_cast<float*>(reinterpret_cast<char*>(data) + (size_t)r * step); }
    const float* ptr(int r) const { return reinterpret_cast<const float*>(reinterpret_cast<const char*>(data) + (size_t)r * step); }
    float& operator()(int r, int c) { return ptr(r)[c]; }
    const float& operator()(int r, int c) const { return ptr(r)[c]; }
    float& operator()(const Point& p) { return ptr(p.y)[p.x]; }
    const float& operator()(const Point& p) const { return ptr(p.y)[p.x]; }
    Mat1f clone() const
    {
        Mat1f m(rows, cols);
        for (int r = 0; r < rows; ++r) std::memcpy(m.ptr(r), ptr(r), (size_t)cols * sizeof(float));
        return m;
    }

private:
    std::shared_ptr<float> owner_;  // shallow copies share storage, like cv::Mat
};

}  // namespace fa
#endif

#endif
