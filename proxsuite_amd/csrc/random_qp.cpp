// Synthetic-input generator of the hot path's benchmark: host-only C++ that
// restates the reference's own random QP families so the CPU baseline and the
// MI355X path are fed the very same problems
//   reference include/proxsuite/proxqp/utils/random_qp_problems.hpp
//     Lehmer-64 RNG + Box-Muller normals            :104-147
//     sparse_positive_definite_rand_not_compressed  :307-334
//     sparse_matrix_rand_not_compressed             :353-368
//     dense_unconstrained_qp                        :438-460
//     dense_strongly_convex_qp                      :462-502
//     dense_not_strongly_convex_qp                  :504-543
//     dense_degenerate_qp                           :545-589
//     dense_box_constrained_qp                      :591-628
// Outputs are row-major fp64 buffers owned by the caller.  The minimal
// eigenvalue (the reference calls Eigen's selfadjoint solver, :329) is computed
// by Householder tridiagonalisation + Sturm bisection, so H agrees with the
// reference's to rounding (~1e-15 relative), not bitwise.
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>
#ifdef _OPENMP
#include <omp.h>
#endif

namespace {

using u64 = std::uint64_t;
using u128 = __uint128_t;

struct Lehmer
{
  u128 state;
  static constexpr u64 K = 0xda942042e4dd58b5ULL;
  Lehmer()
    : state(u128(K) * u128(K))
  {
  }
  u64 next()
  {
    state *= u128(K);
    return u64(state >> 64);
  }
  void set_seed(u64 seed)
  {
    state = u128(seed) + 1;
    next();
    next();
  }
  double uniform()
  {
    u64 a = next() / (1U << 11U);
    return double(a) / double(u64(1) << 53U);
  }
  double normal()
  {
    static const double pi2 = std::atan(1.0) * 8;
    double u1 = uniform();
    double u2 = uniform();
    double ln = std::log(u1);
    double sq = std::sqrt(-2 * ln);
    return sq * std::cos(pi2 * u2);
  }
};

thread_local Lehmer g_rng;

// smallest eigenvalue of the symmetric n x n matrix `a` (row-major, destroyed)
double
min_eigenvalue(std::vector<double>& a, long n)
{
  if (n == 1)
    return a[0];
  std::vector<double> d(static_cast<size_t>(n)), e(static_cast<size_t>(n), 0.0), v(static_cast<size_t>(n)),
    p(static_cast<size_t>(n));
  // Householder tridiagonalisation, trailing-submatrix form
  for (long k = 0; k + 2 < n; ++k) {
    long m = n - k - 1; // size of x = a[k+1.., k]
    double nrm2 = 0;
    for (long i = 0; i < m; ++i) {
      double x = a[size_t((k + 1 + i) * n + k)];
      nrm2 += x * x;
    }
    double nrm = std::sqrt(nrm2);
    double x0 = a[size_t((k + 1) * n + k)];
    if (nrm == 0.0 || nrm2 - x0 * x0 == 0.0) {
      e[size_t(k)] = x0; // already tridiagonal in this column
      continue;
    }
    double alpha = x0 > 0 ? -nrm : nrm;
    double vn2 = 0;
    for (long i = 0; i < m; ++i) {
      double x = a[size_t((k + 1 + i) * n + k)];
      v[size_t(i)] = (i == 0) ? x - alpha : x;
      vn2 += v[size_t(i)] * v[size_t(i)];
    }
    double beta = 2.0 / vn2;
    // p = beta * A22 v ; K = beta/2 v.p ; w = p - K v ; A22 -= v w^T + w v^T
    for (long i = 0; i < m; ++i) {
      double acc = 0;
      const double* row = &a[size_t((k + 1 + i) * n + k + 1)];
      for (long j = 0; j < m; ++j)
        acc += row[j] * v[size_t(j)];
      p[size_t(i)] = beta * acc;
    }
    double vp = 0;
    for (long i = 0; i < m; ++i)
      vp += v[size_t(i)] * p[size_t(i)];
    double Kc = 0.5 * beta * vp;
    for (long i = 0; i < m; ++i)
      p[size_t(i)] -= Kc * v[size_t(i)];
    for (long i = 0; i < m; ++i) {
      double* row = &a[size_t((k + 1 + i) * n + k + 1)];
      double vi = v[size_t(i)], wi = p[size_t(i)];
      for (long j = 0; j < m; ++j)
        row[j] -= vi * p[size_t(j)] + wi * v[size_t(j)];
    }
    e[size_t(k)] = alpha;
  }
  for (long i = 0; i < n; ++i)
    d[size_t(i)] = a[size_t(i * n + i)];
  e[size_t(n - 2)] = a[size_t((n - 1) * n + (n - 2))];
  // Gershgorin bounds then Sturm bisection for the smallest eigenvalue
  double lo = d[0], hi = d[0];
  for (long i = 0; i < n; ++i) {
    double r = (i > 0 ? std::fabs(e[size_t(i - 1)]) : 0.0) + (i + 1 < n ? std::fabs(e[size_t(i)]) : 0.0);
    lo = std::fmin(lo, d[size_t(i)] - r);
    hi = std::fmax(hi, d[size_t(i)] + r);
  }
  auto count_below = [&](double x) {
    long cnt = 0;
    double q = d[0] - x;
    if (q < 0)
      ++cnt;
    for (long i = 1; i < n; ++i) {
      double denom = q;
      if (denom == 0.0)
        denom = 1e-300;
      q = d[size_t(i)] - x - e[size_t(i - 1)] * e[size_t(i - 1)] / denom;
      if (q < 0)
        ++cnt;
    }
    return cnt;
  };
  double a_ = lo, b_ = hi;
  for (int it = 0; it < 200; ++it) {
    double mid = 0.5 * (a_ + b_);
    if (mid == a_ || mid == b_)
      break;
    if (count_below(mid) >= 1)
      b_ = mid;
    else
      a_ = mid;
  }
  return 0.5 * (a_ + b_);
}

// :307-334
void
sparse_positive_definite_rand(Lehmer& rng, long n, double rho, double p, double* H)
{
  std::vector<double> M(size_t(n * n), 0.0);
  for (long i = 0; i < n; ++i)
    for (long j = 0; j < n; ++j) {
      double urandom = rng.uniform();
      if (urandom < p / 2)
        M[size_t(i * n + j)] = rng.normal();
    }
  for (long i = 0; i < n; ++i)
    for (long j = 0; j < n; ++j)
      H[i * n + j] = (M[size_t(i * n + j)] + M[size_t(j * n + i)]) * 0.5;
  std::vector<double> tmp(H, H + n * n);
  double mn = min_eigenvalue(tmp, n);
  for (long i = 0; i < n; ++i)
    H[i * n + i] += rho + std::fabs(mn);
}
// :353-368
void
sparse_matrix_rand(Lehmer& rng, long rows, long cols, double p, double* A)
{
  for (long i = 0; i < rows; ++i)
    for (long j = 0; j < cols; ++j)
      A[i * cols + j] = (rng.uniform() < p) ? rng.normal() : 0.0;
}
void
vector_rand(Lehmer& rng, long n, double* v)
{
  for (long i = 0; i < n; ++i)
    v[i] = rng.normal();
}
void
matvec(const double* A, long rows, long cols, const double* x, double* y)
{
  for (long i = 0; i < rows; ++i) {
    double acc = 0;
    for (long j = 0; j < cols; ++j)
      acc += A[i * cols + j] * x[j];
    y[i] = acc;
  }
}

void
strongly_convex(Lehmer& rng, long n, long n_eq, long n_in, double p, double sc, double* H, double* g,
                double* A, double* b, double* C, double* u, double* l)
{
  sparse_positive_definite_rand(rng, n, sc, p, H);
  vector_rand(rng, n, g);
  sparse_matrix_rand(rng, n_eq, n, p, A);
  sparse_matrix_rand(rng, n_in, n, p, C);
  std::vector<double> x_sol(static_cast<size_t>(n)), delta(static_cast<size_t>(n_in));
  vector_rand(rng, n, x_sol.data());
  for (long i = 0; i < n_in; ++i)
    delta[size_t(i)] = rng.uniform();
  matvec(C, n_in, n, x_sol.data(), u);
  for (long i = 0; i < n_in; ++i) {
    u[i] += delta[size_t(i)];
    l[i] = -1.e20;
  }
  matvec(A, n_eq, n, x_sol.data(), b);
}

} // namespace

extern "C" {

// smallest eigenvalue of a symmetric matrix (Householder tridiagonalisation + Sturm bisection):
// the "ExactMethod" of estimate_minimal_eigen_value_of_symmetric_matrix (reference
// dense/helpers.hpp:160-163 calls Eigen's SelfAdjointEigenSolver there)
double
pqp_min_eigenvalue_symmetric(int64_t n, const double* H)
{
  std::vector<double> tmp(H, H + size_t(n) * size_t(n));
  return min_eigenvalue(tmp, long(n));
}

void
pqp_rand_set_seed(uint64_t seed)
{
  g_rng.set_seed(seed);
}
double
pqp_rand_uniform()
{
  return g_rng.uniform();
}
double
pqp_rand_normal()
{
  return g_rng.normal();
}

// reference random_qp_problems.hpp:462-502 (uses the calling thread's RNG state)
void
pqp_dense_strongly_convex_qp(int64_t n, int64_t n_eq, int64_t n_in, double sparsity_factor,
                             double strong_convexity_factor, double* H, double* g, double* A,
                             double* b, double* C, double* u, double* l)
{
  strongly_convex(g_rng, n, n_eq, n_in, sparsity_factor, strong_convexity_factor, H, g, A, b, C, u, l);
}

// Batch form of the benchmark loop (reference benchmark/timings-parallel.cpp:43-63):
// for i in [seed0, seed0+B): set_seed(i); dense_strongly_convex_qp(...).
// Buffers are [B][...] contiguous.
void
pqp_dense_strongly_convex_qp_batch(int64_t B, uint64_t seed0, int64_t n, int64_t n_eq, int64_t n_in,
                                   double sparsity_factor, double strong_convexity_factor, double* H,
                                   double* g, double* A, double* b, double* C, double* u, double* l)
{
#pragma omp parallel for schedule(dynamic)
  for (int64_t i = 0; i < B; ++i) {
    Lehmer rng;
    rng.set_seed(seed0 + u64(i));
    strongly_convex(rng, n, n_eq, n_in, sparsity_factor, strong_convexity_factor, H + i * n * n,
                    g + i * n, A + i * n_eq * n, b + i * n_eq, C + i * n_in * n, u + i * n_in,
                    l + i * n_in);
  }
}

// :504-543
void
pqp_dense_not_strongly_convex_qp(int64_t n, int64_t n_eq, int64_t n_in, double p, double* H,
                                 double* g, double* A, double* b, double* C, double* u, double* l)
{
  Lehmer& rng = g_rng;
  sparse_positive_definite_rand(rng, n, 0.0, p, H);
  sparse_matrix_rand(rng, n_eq, n, p, A);
  sparse_matrix_rand(rng, n_in, n, p, C);
  std::vector<double> x_sol(static_cast<size_t>(n)), y_sol(static_cast<size_t>(n_eq)), z_sol(static_cast<size_t>(n_in)),
    delta(static_cast<size_t>(n_in)), Cx(static_cast<size_t>(n_in));
  vector_rand(rng, n, x_sol.data());
  vector_rand(rng, n_eq, y_sol.data());
  vector_rand(rng, n_in, z_sol.data());
  for (long i = 0; i < n_in; ++i)
    delta[size_t(i)] = rng.uniform();
  matvec(C, n_in, n, x_sol.data(), Cx.data());
  for (long i = 0; i < n_in; ++i) {
    u[i] = Cx[size_t(i)] + delta[size_t(i)];
    l[i] = Cx[size_t(i)] - delta[size_t(i)];
  }
  matvec(A, n_eq, n, x_sol.data(), b);
  for (long k = 0; k < n; ++k) {
    double acc = 0;
    for (long j = 0; j < n; ++j)
      acc += H[k * n + j] * x_sol[size_t(j)];
    for (long i = 0; i < n_in; ++i)
      acc += C[i * n + k] * z_sol[size_t(i)];
    for (long i = 0; i < n_eq; ++i)
      acc += A[i * n + k] * y_sol[size_t(i)];
    g[k] = -acc;
  }
}

// :545-589 -- C, u, l have 2*n_in rows
void
pqp_dense_degenerate_qp(int64_t n, int64_t n_eq, int64_t n_in, double p, double sc, double* H,
                        double* g, double* A, double* b, double* C, double* u, double* l)
{
  Lehmer& rng = g_rng;
  sparse_positive_definite_rand(rng, n, sc, p, H);
  vector_rand(rng, n, g);
  sparse_matrix_rand(rng, n_eq, n, p, A);
  std::vector<double> x_sol(static_cast<size_t>(n)), delta(size_t(2 * n_in));
  vector_rand(rng, n, x_sol.data());
  for (long i = 0; i < 2 * n_in; ++i)
    delta[size_t(i)] = rng.uniform();
  matvec(A, n_eq, n, x_sol.data(), b);
  sparse_matrix_rand(rng, n_in, n, p, C);
  std::memcpy(C + n_in * n, C, size_t(n_in * n) * sizeof(double));
  matvec(C, 2 * n_in, n, x_sol.data(), u);
  for (long i = 0; i < 2 * n_in; ++i) {
    u[i] += delta[size_t(i)];
    l[i] = -1.e20;
  }
}

// :591-628 -- C = identity pattern (n_in x n)
void
pqp_dense_box_constrained_qp(int64_t n, int64_t n_eq, int64_t n_in, double p, double sc, double* H,
                             double* g, double* A, double* b, double* C, double* u, double* l)
{
  Lehmer& rng = g_rng;
  sparse_positive_definite_rand(rng, n, sc, p, H);
  vector_rand(rng, n, g);
  sparse_matrix_rand(rng, n_eq, n, p, A);
  std::vector<double> x_sol(static_cast<size_t>(n)), delta(static_cast<size_t>(n_in));
  vector_rand(rng, n, x_sol.data());
  for (long i = 0; i < n_in; ++i)
    delta[size_t(i)] = rng.uniform();
  matvec(A, n_eq, n, x_sol.data(), b);
  std::memset(C, 0, size_t(n_in * n) * sizeof(double));
  for (long i = 0; i < (n_in < n ? n_in : n); ++i)
    C[i * n + i] += 1;
  for (long i = 0; i < n_in; ++i) {
    u[i] = x_sol[size_t(i)] + delta[size_t(i)];
    l[i] = x_sol[size_t(i)] - delta[size_t(i)];
  }
}

// :438-460
void
pqp_dense_unconstrained_qp(int64_t n, double p, double sc, double* H, double* g)
{
  Lehmer& rng = g_rng;
  sparse_positive_definite_rand(rng, n, sc, p, H);
  vector_rand(rng, n, g);
}

} // extern "C"
