// proxsuite/proxqp/utils/random_qp_problems.hpp -- the reference's synthetic QP generators
// (include/proxsuite/proxqp/utils/random_qp_problems.hpp:104-147 RNG, :462-628 families) as a
// thin C++ layer over libpqp_randqp.so (proxsuite_amd/csrc/random_qp.cpp), which reproduces the
// reference's 128-bit Lehmer generator and draw order so that `set_seed(i)` yields the
// benchmark's QP number i (benchmark/timings-parallel.cpp:43-63).
#ifndef PROXSUITE_AMD_PROXQP_UTILS_RANDOM_QP_PROBLEMS_HPP
#define PROXSUITE_AMD_PROXQP_UTILS_RANDOM_QP_PROBLEMS_HPP

#include <cstdint>

#include "proxsuite/proxqp/dense/model.hpp"

extern "C" {
void pqp_rand_set_seed(uint64_t seed);
double pqp_rand_uniform();
double pqp_rand_normal();
void pqp_dense_strongly_convex_qp(int64_t n, int64_t n_eq, int64_t n_in, double sparsity_factor,
                                  double strong_convexity_factor, double* H, double* g, double* A, double* b,
                                  double* C, double* u, double* l);
void pqp_dense_not_strongly_convex_qp(int64_t n, int64_t n_eq, int64_t n_in, double p, double* H, double* g,
                                      double* A, double* b, double* C, double* u, double* l);
void pqp_dense_box_constrained_qp(int64_t n, int64_t n_eq, int64_t n_in, double p, double sc, double* H,
                                  double* g, double* A, double* b, double* C, double* u, double* l);
}

namespace proxsuite {
namespace proxqp {
namespace utils {
namespace rand {
inline void
set_seed(uint64_t seed)
{
  pqp_rand_set_seed(seed);
}
inline double
uniform_rand()
{
  return pqp_rand_uniform();
}
inline double
normal_rand()
{
  return pqp_rand_normal();
}
// reference utils/random_qp_problems.hpp:150-175: normal draws, in index order
template<typename Scalar>
dense::Vec<Scalar>
vector_rand(isize nrows)
{
  dense::Vec<Scalar> v(nrows);
  for (isize i = 0; i < nrows; ++i)
    v(i) = Scalar(normal_rand());
  return v;
}
template<typename Scalar>
dense::Mat<Scalar>
matrix_rand(isize nrows, isize ncols)
{
  dense::Mat<Scalar> m(nrows, ncols);
  for (isize i = 0; i < nrows; ++i)
    for (isize j = 0; j < ncols; ++j)
      m(i, j) = Scalar(normal_rand());
  return m;
}
} // namespace rand

template<typename T = double>
dense::Model<T>
dense_strongly_convex_qp(isize dim, isize n_eq, isize n_in, T sparsity_factor, T strong_convexity_factor = T(1e-2))
{
  dense::Model<T> m(dim, n_eq, n_in);
  pqp_dense_strongly_convex_qp(dim, n_eq, n_in, sparsity_factor, strong_convexity_factor, m.H.data(),
                               m.g.data(), m.A.data(), m.b.data(), m.C.data(), m.u.data(), m.l.data());
  return m;
}

template<typename T = double>
dense::Model<T>
dense_not_strongly_convex_qp(isize dim, isize n_eq, isize n_in, T sparsity_factor)
{
  dense::Model<T> m(dim, n_eq, n_in);
  pqp_dense_not_strongly_convex_qp(dim, n_eq, n_in, sparsity_factor, m.H.data(), m.g.data(), m.A.data(),
                                   m.b.data(), m.C.data(), m.u.data(), m.l.data());
  return m;
}

template<typename T = double>
dense::Model<T>
dense_box_constrained_qp(isize dim, isize n_eq, isize n_in, T sparsity_factor, T strong_convexity_factor = T(1e-2))
{
  dense::Model<T> m(dim, n_eq, n_in);
  pqp_dense_box_constrained_qp(dim, n_eq, n_in, sparsity_factor, strong_convexity_factor, m.H.data(),
                               m.g.data(), m.A.data(), m.b.data(), m.C.data(), m.u.data(), m.l.data());
  return m;
}

} // namespace utils
} // namespace proxqp
} // namespace proxsuite

#endif
