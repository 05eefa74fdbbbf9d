// TEST INFRASTRUCTURE ONLY -- plain-C entry points of the CPU oracle so that
// tests/ (ctypes) and bench.py's cpu_baseline leg can drive it.  Nothing in the
// product path may link or load this library.
#include "proxqp_oracle.hpp"

#include <algorithm>
#include <cstdio>
#include <cstring>
#ifdef _OPENMP
#include <omp.h>
#include <sched.h>
#include <vector>
#endif

using pqo::isize;
using pqo::QP;

namespace {
thread_local std::string g_last_error;
const double*
opt(const double* p)
{
  return p;
}
}

extern "C" {

const char*
pqo_last_error()
{
  return g_last_error.c_str();
}

void*
pqo_create(int64_t n, int64_t n_eq, int64_t n_in, int box, int hessian_type, int dense_backend)
{
  try {
    return new QP(n, n_eq, n_in, box != 0, hessian_type, dense_backend);
  } catch (std::exception& e) {
    g_last_error = e.what();
    return nullptr;
  }
}

void
pqo_destroy(void* h)
{
  delete static_cast<QP*>(h);
}

pqp_settings*
pqo_settings(void* h)
{
  return &static_cast<QP*>(h)->settings;
}

pqp_info*
pqo_info(void* h)
{
  return &static_cast<QP*>(h)->results.info;
}

// settings.verbose: the per-iteration records of the last solve (proxqp_oracle.hpp, QP::trace), 8 doubles each;
// returns their number, copies at most `cap` of them
int64_t
pqo_trace(void* h, double* out, int64_t cap)
{
  const std::vector<double>& t = static_cast<QP*>(h)->trace;
  const int64_t n = int64_t(t.size() / 8);
  if (out)
    std::copy(t.begin(), t.begin() + std::min(n, cap) * 8, out);
  return n;
}

int
pqo_dense_backend(void* h)
{
  return static_cast<QP*>(h)->dense_backend;
}

int
pqo_init(void* h, const double* H, const double* g, const double* A, const double* b,
         const double* C, const double* l, const double* u, const double* l_box,
         const double* u_box, int compute_preconditioner, double rho, double mu_eq, double mu_in,
         double manual_minimal_H_eigenvalue)
{
  static_cast<QP*>(h)->init(opt(H), g, A, b, C, l, u, l_box, u_box, compute_preconditioner != 0,
                            rho, mu_eq, mu_in, manual_minimal_H_eigenvalue);
  return 0;
}

int
pqo_update(void* h, const double* H, const double* g, const double* A, const double* b,
           const double* C, const double* l, const double* u, const double* l_box,
           const double* u_box, int update_preconditioner, double rho, double mu_eq, double mu_in,
           double manual_minimal_H_eigenvalue)
{
  static_cast<QP*>(h)->update(H, g, A, b, C, l, u, l_box, u_box, update_preconditioner != 0, rho,
                              mu_eq, mu_in, manual_minimal_H_eigenvalue);
  return 0;
}

int
pqo_solve(void* h, const double* x, const double* y, const double* z)
{
  static_cast<QP*>(h)->solve(x, y, z);
  return 0;
}

// dense/compute_ECJ.hpp:29-189; outputs may be NULL
int
pqo_compute_backward(void* h, const double* loss_derivative, double eps, double rho_new, double mu_new,
                     double* dL_dH, double* dL_dg, double* dL_dA, double* dL_db, double* dL_dC, double* dL_du,
                     double* dL_dl)
{
  QP* q = static_cast<QP*>(h);
  try {
    q->compute_backward(loss_derivative, eps, rho_new, mu_new);
  } catch (const std::exception& e) {
    g_last_error = e.what();
    return -1;
  }
  auto cp = [](double* dst, const pqo::Vec& v) {
    if (dst && !v.empty())
      std::memcpy(dst, v.data(), v.size() * sizeof(double));
  };
  const pqo::BackwardData& b = q->backward_data;
  cp(dL_dH, b.dL_dH);
  cp(dL_dg, b.dL_dg);
  cp(dL_dA, b.dL_dA);
  cp(dL_db, b.dL_db);
  cp(dL_dC, b.dL_dC);
  cp(dL_du, b.dL_du);
  cp(dL_dl, b.dL_dl);
  return 0;
}

void
pqo_cleanup(void* h)
{
  static_cast<QP*>(h)->cleanup();
}

void
pqo_get_results(void* h, double* x, double* y, double* z, double* se, double* si, pqp_info* info)
{
  QP* q = static_cast<QP*>(h);
  auto cp = [](double* dst, const pqo::Vec& v) {
    if (dst && !v.empty())
      std::memcpy(dst, v.data(), v.size() * sizeof(double));
  };
  cp(x, q->results.x);
  cp(y, q->results.y);
  cp(z, q->results.z);
  cp(se, q->results.se);
  cp(si, q->results.si);
  if (info)
    *info = q->results.info;
}

// scaled data & preconditioner, for stage-by-stage comparison with the device
void
pqo_get_scaled(void* h, double* H, double* g, double* A, double* b, double* C, double* l,
               double* u, double* delta, double* c)
{
  QP* q = static_cast<QP*>(h);
  auto cp = [](double* dst, const pqo::Vec& v) {
    if (dst && !v.empty())
      std::memcpy(dst, v.data(), v.size() * sizeof(double));
  };
  cp(H, q->work.H_scaled);
  cp(g, q->work.g_scaled);
  cp(A, q->work.A_scaled);
  cp(b, q->work.b_scaled);
  cp(C, q->work.C_scaled);
  cp(l, q->work.l_scaled);
  cp(u, q->work.u_scaled);
  cp(delta, q->ruiz.delta);
  if (c)
    *c = q->ruiz.c;
}

// counters[0..] = fact_flops, fact_bytes, level2_flops, n_solves, n_residuals,
// n_ls_evals, n_inserted, n_deleted, n_refactorize   
void
pqo_get_counters(void* h, double* out)
{
  QP* q = static_cast<QP*>(h);
  out[0] = q->counters.fact_flops;
  out[1] = q->counters.fact_bytes;
  out[2] = q->counters.level2_flops;
  out[3] = double(q->n_solves);
  out[4] = double(q->n_residuals);
  out[5] = double(q->n_ls_evals);
  out[6] = double(q->n_inserted);
  out[7] = double(q->n_deleted);
  out[8] = double(q->n_refactorize);
}

// The reference's data-parallel driver, restated:
// include/proxsuite/proxqp/parallel/qp_solve.hpp:41-59
// (`#pragma omp parallel for schedule(dynamic)` over independent QPs).
//
// The team's threads are bound to distinct CPUs of the caller's affinity mask for the duration of
// the loop (what OMP_PROC_BIND=spread does, without depending on the environment): freshly created
// OpenMP workers otherwise start on the caller's CPU and some kernels take about a second to spread
// them, during which the "parallel" loop time-shares one core (measured: 8 threads no faster than
// 1 for the first five calls).  As a timed CPU baseline that would flatter the GPU.
int
pqo_solve_in_parallel(void** handles, int64_t count, int num_threads)
{
#ifdef _OPENMP
  int nt = num_threads > 0 ? num_threads : std::max(omp_get_max_threads() / 2, 1);
  omp_set_dynamic(0);
  omp_set_num_threads(nt);
  cpu_set_t caller_mask;
  std::vector<int> cpus;
  const bool have_mask = sched_getaffinity(0, sizeof(caller_mask), &caller_mask) == 0;
  if (have_mask)
    for (int c = 0; c < CPU_SETSIZE; ++c)
      if (CPU_ISSET(c, &caller_mask))
        cpus.push_back(c);
#pragma omp parallel
  {
    if (nt > 1 && !cpus.empty()) {
      cpu_set_t one;
      CPU_ZERO(&one);
      CPU_SET(cpus[size_t(omp_get_thread_num()) % cpus.size()], &one);
      sched_setaffinity(0, sizeof(one), &one);
    }
#pragma omp for schedule(dynamic)
    for (int64_t i = 0; i < count; ++i)
      static_cast<QP*>(handles[i])->solve(nullptr, nullptr, nullptr);
    // the caller gets its mask back; the pool's workers stay where they are (as with OMP_PROC_BIND)
    if (nt > 1 && have_mask && omp_get_thread_num() == 0)
      sched_setaffinity(0, sizeof(caller_mask), &caller_mask);
  }
  return nt;
#else
  (void)num_threads;
  for (int64_t i = 0; i < count; ++i)
    static_cast<QP*>(handles[i])->solve(nullptr, nullptr, nullptr);
  return 1;
#endif
}

int
pqo_max_threads()
{
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}

} // extern "C"
