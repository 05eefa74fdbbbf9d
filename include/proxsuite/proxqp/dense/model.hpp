// proxsuite/proxqp/dense/model.hpp -- Model<T>: host copy of the QP data handed to
// QP::init / QP::update (reference include/proxsuite/proxqp/dense/model.hpp:24-147).  The device
// keeps its own (scaled and unscaled) copies; this one exists because `qp.model` is a public
// member users read.
#ifndef PROXSUITE_AMD_PROXQP_DENSE_MODEL_HPP
#define PROXSUITE_AMD_PROXQP_DENSE_MODEL_HPP

#include <stdexcept>

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

  // reference model.hpp:126-145 (symmetry of H; sizes are fixed by construction here)
  bool is_valid(bool /*box_constraints*/ = false) const
  {
    for (isize i = 0; i < dim; ++i)
      for (isize j = 0; j < i; ++j)
        if (std::fabs(H(i, j) - H(j, i)) > T(1e-12) * (T(1) + std::fabs(H(i, j))))
          return false;
    return true;
  }
};

} // namespace dense
} // namespace proxqp
} // namespace proxsuite

#endif
