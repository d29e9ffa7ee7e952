// Matrix.h -- the few lines of a dense, column-major, dynamically sized double matrix that the host
// side needs in its signatures.  In the reference tree this name is Eigen::MatrixXd
// (include/StdInclude.h:7); Eigen is not available in this image, so the shim carries its own type with
// the same storage order and the same accessor spelling.  INTEGRATION.md shows the one-line mapping.
#pragma once
#include <cstddef>
#include <vector>

class MatrixXd {
public:
    MatrixXd() : r_(0), c_(0) {}
    MatrixXd(int rows, int cols) : r_(rows), c_(cols), d_((size_t)rows * cols, 0.0) {}
    int rows() const { return r_; }
    int cols() const { return c_; }
    int size() const { return r_ * c_; }
    void resize(int rows, int cols) { r_ = rows; c_ = cols; d_.assign((size_t)rows * cols, 0.0); }
    void setZero() { d_.assign(d_.size(), 0.0); }
    double &operator()(int i, int j) { return d_[(size_t)i + (size_t)j * r_]; }
    double operator()(int i, int j) const { return d_[(size_t)i + (size_t)j * r_]; }
    double &operator()(int i) { return d_[(size_t)i]; }           // vectors
    double operator()(int i) const { return d_[(size_t)i]; }
    double *data() { return d_.data(); }
    const double *data() const { return d_.data(); }
private:
    int r_, c_;
    std::vector<double> d_;
};
