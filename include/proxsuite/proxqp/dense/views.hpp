// proxsuite/proxqp/dense/views.hpp -- the small matrix / vector vocabulary of the MI355X
// dense ProxQP facade.
//
// The reference takes Eigen::Ref arguments (include/proxsuite/proxqp/dense/fwd.hpp:16-31:
// row-major Mat<T>, column Vec<T>, MatRef / VecRef).  Eigen is not a dependency of this build,
// so the same four names are provided here:
//   Vec<T>, Mat<T>      owning, contiguous, row-major (the layout the C-ABI consumes);
//   VecRef<T>, MatRef<T> non-owning strided views, implicitly constructible from Vec / Mat,
//                        from raw pointers, and from any Eigen-like dense object (anything
//                        with data(), rows(), cols(), innerStride(), outerStride() and
//                        IsRowMajor) -- so code written against Eigen keeps compiling when
//                        Eigen is on the include path.
#ifndef PROXSUITE_AMD_PROXQP_DENSE_VIEWS_HPP
#define PROXSUITE_AMD_PROXQP_DENSE_VIEWS_HPP

#include <algorithm>
#include <cmath>
#include <ostream>
#include <type_traits>
#include <vector>

#include "proxsuite/proxqp/settings.hpp"

namespace proxsuite {
namespace proxqp {
namespace dense {

using proxsuite::proxqp::isize; // (the reference re-exports veg's isize / usize in this namespace: dense/fwd.hpp)
using proxsuite::proxqp::usize;

namespace detail {
// `v.array() += c` and friends: the coefficient-wise view the reference's examples use on Eigen objects
// (examples/cpp/init_dense_qp_with_box.cpp:24-25)
template<typename T>
struct ArrayProxy
{
  T* p;
  isize n;
  ArrayProxy& operator+=(T c)
  {
    for (isize i = 0; i < n; ++i)
      p[i] += c;
    return *this;
  }
  ArrayProxy& operator-=(T c)
  {
    for (isize i = 0; i < n; ++i)
      p[i] -= c;
    return *this;
  }
  ArrayProxy& operator*=(T c)
  {
    for (isize i = 0; i < n; ++i)
      p[i] *= c;
    return *this;
  }
  ArrayProxy& operator/=(T c)
  {
    for (isize i = 0; i < n; ++i)
      p[i] /= c;
    return *this;
  }
};
} // namespace detail

template<typename T>
class Vec
{
public:
  Vec() = default;
  explicit Vec(isize n, T value = T(0))
    : v_(usize(n), value)
  {
  }
  Vec(std::initializer_list<T> il)
    : v_(il)
  {
  }
  isize size() const { return isize(v_.size()); }
  isize rows() const { return size(); }
  isize cols() const { return 1; }
  T* data() { return v_.data(); }
  const T* data() const { return v_.data(); }
  T& operator[](isize i) { return v_[usize(i)]; }
  const T& operator[](isize i) const { return v_[usize(i)]; }
  T& operator()(isize i) { return v_[usize(i)]; }
  const T& operator()(isize i) const { return v_[usize(i)]; }
  void resize(isize n) { v_.assign(usize(n), T(0)); }
  void setZero() { std::fill(v_.begin(), v_.end(), T(0)); }
  void setConstant(T c) { std::fill(v_.begin(), v_.end(), c); }
  void setOnes() { setConstant(T(1)); }
  detail::ArrayProxy<T> array() { return { v_.data(), size() }; }
  T* begin() { return v_.data(); }
  T* end() { return v_.data() + v_.size(); }
  const T* begin() const { return v_.data(); }
  const T* end() const { return v_.data() + v_.size(); }

private:
  std::vector<T> v_;
};

template<typename T>
class Mat
{
public:
  Mat() = default;
  Mat(isize rows, isize cols, T value = T(0))
    : r_(rows)
    , c_(cols)
    , v_(usize(rows * cols), value)
  {
  }
  isize rows() const { return r_; }
  isize cols() const { return c_; }
  isize size() const { return r_ * c_; }
  T* data() { return v_.data(); }
  const T* data() const { return v_.data(); }
  T& operator()(isize i, isize j) { return v_[usize(i * c_ + j)]; }
  const T& operator()(isize i, isize j) const { return v_[usize(i * c_ + j)]; }
  void resize(isize rows, isize cols)
  {
    r_ = rows;
    c_ = cols;
    v_.assign(usize(rows * cols), T(0));
  }
  void setZero() { std::fill(v_.begin(), v_.end(), T(0)); }
  void setIdentity()
  {
    setZero();
    for (isize i = 0; i < std::min(r_, c_); ++i)
      (*this)(i, i) = T(1);
  }
  detail::ArrayProxy<T> array() { return { v_.data(), size() }; }

private:
  isize r_ = 0, c_ = 0;
  std::vector<T> v_;
};

namespace detail {
template<typename X, typename = void>
struct is_eigen_like : std::false_type
{};
template<typename X>
struct is_eigen_like<X,
                     std::void_t<decltype(std::declval<const X&>().data()),
                                 decltype(std::declval<const X&>().rows()),
                                 decltype(std::declval<const X&>().cols()),
                                 decltype(std::declval<const X&>().innerStride()),
                                 decltype(std::declval<const X&>().outerStride()),
                                 decltype(X::IsRowMajor)>> : std::true_type
{};
} // namespace detail

template<typename T>
struct VecRef
{
  const T* ptr = nullptr;
  isize n = 0;
  isize stride = 1;
  VecRef() = default;
  VecRef(const T* p, isize size, isize inc = 1)
    : ptr(p)
    , n(size)
    , stride(inc)
  {
  }
  VecRef(const Vec<T>& v)
    : ptr(v.data())
    , n(v.size())
  {
  }
  VecRef(const std::vector<T>& v)
    : ptr(v.data())
    , n(isize(v.size()))
  {
  }
  template<typename X, typename = std::enable_if_t<detail::is_eigen_like<X>::value>>
  VecRef(const X& x)
    : ptr(x.data())
    , n(isize(x.rows() * x.cols()))
    , stride(isize(x.innerStride()))
  {
  }
  isize rows() const { return n; }
  isize size() const { return n; }
  const T& operator[](isize i) const { return ptr[i * stride]; }
};

template<typename T>
struct MatRef
{
  const T* ptr = nullptr;
  isize r = 0, c = 0;
  isize row_stride = 0, col_stride = 1;
  MatRef() = default;
  // row-major contiguous by default
  MatRef(const T* p, isize rows, isize cols)
    : ptr(p)
    , r(rows)
    , c(cols)
    , row_stride(cols)
  {
  }
  MatRef(const T* p, isize rows, isize cols, isize rs, isize cs)
    : ptr(p)
    , r(rows)
    , c(cols)
    , row_stride(rs)
    , col_stride(cs)
  {
  }
  MatRef(const Mat<T>& m)
    : ptr(m.data())
    , r(m.rows())
    , c(m.cols())
    , row_stride(m.cols())
  {
  }
  template<typename X, typename = std::enable_if_t<detail::is_eigen_like<X>::value>>
  MatRef(const X& x)
    : ptr(x.data())
    , r(isize(x.rows()))
    , c(isize(x.cols()))
    , row_stride(X::IsRowMajor ? isize(x.outerStride()) : isize(x.innerStride()))
    , col_stride(X::IsRowMajor ? isize(x.innerStride()) : isize(x.outerStride()))
  {
  }
  isize rows() const { return r; }
  isize cols() const { return c; }
  isize size() const { return r * c; }
  const T& operator()(isize i, isize j) const { return ptr[i * row_stride + j * col_stride]; }
  bool is_packed_row_major() const { return col_stride == 1 && (row_stride == c || r <= 1); }
};

// the little vector arithmetic the reference's examples do on model data (examples/cpp/update_dense_qp_ws_previous_result.cpp:36:
// `qp_random.g * 0.95`)
template<typename T>
Vec<T>
operator*(const Vec<T>& v, T c)
{
  Vec<T> r(v.size());
  for (isize i = 0; i < v.size(); ++i)
    r[i] = v[i] * c;
  return r;
}
template<typename T>
Vec<T>
operator*(T c, const Vec<T>& v)
{
  return v * c;
}
template<typename T>
Vec<T>
operator+(const Vec<T>& a, const Vec<T>& b)
{
  Vec<T> r(a.size());
  for (isize i = 0; i < a.size(); ++i)
    r[i] = a[i] + b[i];
  return r;
}
template<typename T>
Vec<T>
operator-(const Vec<T>& a, const Vec<T>& b)
{
  Vec<T> r(a.size());
  for (isize i = 0; i < a.size(); ++i)
    r[i] = a[i] - b[i];
  return r;
}

// printing (the examples stream results.x / y / z): one coefficient per line for a vector, one row per line for a matrix
template<typename T>
std::ostream&
operator<<(std::ostream& os, const Vec<T>& v)
{
  for (isize i = 0; i < v.size(); ++i)
    os << (i ? "\n" : "") << v[i];
  return os;
}
template<typename T>
std::ostream&
operator<<(std::ostream& os, const Mat<T>& m)
{
  for (isize i = 0; i < m.rows(); ++i) {
    for (isize j = 0; j < m.cols(); ++j)
      os << (j ? " " : "") << m(i, j);
    if (i + 1 < m.rows())
      os << "\n";
  }
  return os;
}

// |v|_inf, the norm every acceptance test of the reference uses
template<typename T>
T
infty_norm(const Vec<T>& v)
{
  T m = 0;
  for (isize i = 0; i < v.size(); ++i)
    m = std::max(m, std::fabs(v[i]));
  return m;
}

} // namespace dense
} // namespace proxqp
} // namespace proxsuite

#endif
