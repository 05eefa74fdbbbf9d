// proxsuite/proxqp/dense/model.hpp -- Model<T>: host copy of the QP data handed to
// QP::init / QP::update (reference include/proxsuite/proxqp/dense/model.hpp:24-147).  The device
// keeps its own (scaled and unscaled) copies; this one exists because `qp.model` is a public
// member users read.
#ifndef PROXSUITE_AMD_PROXQP_DENSE_MODEL_HPP
#define PROXSUITE_AMD_PROXQP_DENSE_MODEL_HPP

#include <cmath>
#include <limits>
#include <stdexcept>
#include <string>

#include "proxsuite/proxqp/dense/views.hpp"

namespace proxsuite {
namespace proxqp {
namespace dense {

// jacobians of a loss wrt the model, filled by dense::compute_backward
// (reference include/proxsuite/proxqp/dense/backward_data.hpp:27-133)
template<typename T>
struct BackwardData
{
  Mat<T> dL_dH;
  Vec<T> dL_dg;
  Mat<T> dL_dA;
  Vec<T> dL_db;
  Mat<T> dL_dC;
  Vec<T> dL_du;
  Vec<T> dL_dl;
  void initialize(isize dim, isize n_eq, isize n_in)
  {
    dL_dH.resize(dim, dim);
    dL_dg.resize(dim);
    dL_dA.resize(n_eq, dim);
    dL_db.resize(n_eq);
    dL_dC.resize(n_in, dim);
    dL_du.resize(n_in);
    dL_dl.resize(n_in);
  }
};

template<typename T>
struct Model
{
  BackwardData<T> backward_data;
  Mat<T> H;
  Vec<T> g;
  Mat<T> A;
  Mat<T> C;
  Vec<T> b, u, l;
  Vec<T> u_box, l_box;
  isize dim = 0, n_eq = 0, n_in = 0, n_total = 0;

  Model() = default;
  Model(isize dim_, isize n_eq_, isize n_in_, bool box_constraints = false)
    : H(dim_, dim_)
    , g(dim_)
    , A(n_eq_, dim_)
    , C(n_in_, dim_)
    , b(n_eq_)
    , u(n_in_)
    , l(n_in_)
    , dim(dim_)
    , n_eq(n_eq_)
    , n_in(n_in_)
    , n_total(dim_ + n_eq_ + n_in_)
  {
    if (dim_ == 0) // reference model.hpp:65-68
      throw std::invalid_argument(
        "wrong argument size: the dimension wrt the primal variable x should be strictly positive.");
    if (box_constraints) {
      u_box = Vec<T>(dim_, T(1e20));
      l_box = Vec<T>(dim_, T(-1e20));
    }
  }

  // reference model.hpp:104-148: sizes, symmetry of H to machine precision (Eigen's isApprox: ||H - H^T||_F <= eps ||H||_F),
  // C not identically zero when there are inequality rows; std::invalid_argument with the reference's messages
  bool is_valid(bool box_constraints = false) const
  {
    auto size_is = [](isize got, isize want, const char* what) {
      if (got != want)
        throw std::invalid_argument(std::string("wrong argument size: expected ") + std::to_string(want) + ", got " +
                                    std::to_string(got) + "\n" + what);
    };
    size_is(g.size(), dim, "g has not the expected size.");
    size_is(b.size(), n_eq, "b has not the expected size.");
    size_is(l.size(), n_in, "l has not the expected size.");
    size_is(u.size(), n_in, "u has not the expected size.");
    if (box_constraints) {
      size_is(u_box.size(), dim, "u_box has not the expected size");
      size_is(l_box.size(), dim, "l_box has not the expected size");
    }
    if (H.rows() * H.cols() != 0) {
      size_is(H.rows(), dim, "H has not the expected number of rows.");
      size_is(H.cols(), dim, "H has not the expected number of cols.");
      T diff2 = T(0), norm2 = T(0);
      for (isize i = 0; i < dim; ++i)
        for (isize j = 0; j < dim; ++j) {
          const T e = H(i, j) - H(j, i);
          diff2 += e * e;
          norm2 += H(i, j) * H(i, j);
        }
      const T eps = std::numeric_limits<T>::epsilon();
      if (diff2 > eps * eps * norm2)
        throw std::invalid_argument("H is not symmetric.");
    }
    if (A.rows() * A.cols() != 0) {
      size_is(A.rows(), n_eq, "A has not the expected number of rows.");
      size_is(A.cols(), dim, "A has not the expected number of cols.");
    }
    if (C.rows() * C.cols() != 0) {
      size_is(C.rows(), n_in, "C has not the expected number of rows.");
      size_is(C.cols(), dim, "C has not the expected number of cols.");
      bool zero = true;
      for (isize i = 0; i < C.rows() && zero; ++i)
        for (isize j = 0; j < C.cols(); ++j)
          if (std::fabs(C(i, j)) > T(1e-12)) {
            zero = false;
            break;
          }
      if (zero)
        throw std::invalid_argument("C is zero, while n_in != 0.");
    }
    return true;
  }
};

} // namespace dense
} // namespace proxqp
} // namespace proxsuite

#endif
