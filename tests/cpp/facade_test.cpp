// Exercises the C++17 facade (include/proxsuite/...) the way the reference's own programs do:
// benchmark/timings-parallel.cpp:178-220 (BatchQP + solve_in_parallel), test/src/
// dense_qp_wrapper.cpp (init / update / warm start), test/src/parallel_qp_solve.cpp:33-76
// (parallel == serial), test/src/dense_qp_solve.cpp (one-shot solve).  Linked against
// libproxqp_hip.so on the GPU box and against the SIMT-emulator build of the same device
// sources (tests/emu/libpqp_emu.so) in the CPU test-suite.  Exit code 0 = all checks passed.
#include <cstdio>
#include <cstdlib>
#include <iostream>
#include <thread>

#include <proxsuite/proxqp/dense/dense.hpp>
#include <proxsuite/proxqp/parallel/qp_solve.hpp>
#include <proxsuite/proxqp/utils/random_qp_problems.hpp>

using namespace proxsuite::proxqp;
using T = double;

static int failures = 0;
#define CHECK(cond)                                                                                  \
  do {                                                                                               \
    if (!(cond)) {                                                                                   \
      std::printf("CHECK failed %s:%d: %s\n", __FILE__, __LINE__, #cond);                            \
      ++failures;                                                                                    \
    }                                                                                                \
  } while (0)

// unscaled residuals, as every reference test computes them (dense_qp_with_eq_and_in.cpp:46-56)
static void
residuals(const dense::Model<T>& m, const Results<T>& r, T& pri, T& dua)
{
  pri = 0;
  dua = 0;
  for (isize i = 0; i < m.n_eq; ++i) {
    T s = -m.b[i];
    for (isize j = 0; j < m.dim; ++j)
      s += m.A(i, j) * r.x[j];
    pri = std::max(pri, std::fabs(s));
  }
  for (isize i = 0; i < m.n_in; ++i) {
    T s = 0;
    for (isize j = 0; j < m.dim; ++j)
      s += m.C(i, j) * r.x[j];
    pri = std::max(pri, std::max(T(0), s - m.u[i]) + std::max(T(0), m.l[i] - s));
  }
  for (isize j = 0; j < m.dim; ++j) {
    T s = m.g[j];
    for (isize k = 0; k < m.dim; ++k)
      s += m.H(j, k) * r.x[k];
    for (isize i = 0; i < m.n_eq; ++i)
      s += m.A(i, j) * r.y[i];
    for (isize i = 0; i < m.n_in; ++i)
      s += m.C(i, j) * r.z[i];
    dua = std::max(dua, std::fabs(s));
  }
}

int
main(int argc, char** argv)
{
  const isize dim = 10, n_eq = 3, n_in = 4;
  const int num_qps = argc > 1 ? std::atoi(argv[1]) : 6;
  const T eps_abs = 1e-9, sparsity_factor = 0.15, strong_convexity_factor = 1e-2;

  // --- timings-parallel.cpp:178-220: BatchQP, init in place, solve in parallel
  dense::BatchQP<T> qps_vector = dense::BatchQP<T>(usize(num_qps));
  std::vector<dense::Model<T>> models;
  for (int i = 0; i < num_qps; i++) {
    utils::rand::set_seed(uint64_t(i));
    dense::Model<T> qp_random = utils::dense_strongly_convex_qp(dim, n_eq, n_in, sparsity_factor, strong_convexity_factor);
    auto& qp = qps_vector.init_qp_in_place(dim, n_eq, n_in);
    qp.settings.eps_abs = eps_abs;
    qp.settings.eps_rel = 0;
    qp.settings.initial_guess = InitialGuessStatus::NO_INITIAL_GUESS;
    qp.init(qp_random.H, qp_random.g, qp_random.A, qp_random.b, qp_random.C, qp_random.l, qp_random.u);
    CHECK(qp.results.info.status == QPSolverOutput::PROXQP_MAX_ITER_REACHED); // preset by setup (results.hpp:172)
    models.push_back(qp_random);
  }
  CHECK(qps_vector.size() == num_qps);
  dense::solve_in_parallel(qps_vector, 4);
  for (int i = 0; i < num_qps; i++) {
    T pri, dua;
    residuals(models[usize(i)], qps_vector[i].results, pri, dua);
    CHECK(qps_vector[i].results.info.status == QPSolverOutput::PROXQP_SOLVED);
    CHECK(pri <= eps_abs);
    CHECK(dua <= eps_abs);
  }

  // --- parallel_qp_solve.cpp:33-76: a std::vector of QPs solved one by one gives the same x
  std::vector<dense::QP<T>> qps;
  for (int i = 0; i < num_qps; i++) {
    dense::QP<T> qp{ dim, n_eq, n_in };
    qp.settings.eps_abs = eps_abs;
    qp.settings.eps_rel = 0;
    qp.settings.initial_guess = InitialGuessStatus::NO_INITIAL_GUESS;
    const auto& m = models[usize(i)];
    qp.init(m.H, m.g, m.A, m.b, m.C, m.l, m.u);
    qps.push_back(std::move(qp));
  }
  dense::solve_in_parallel(qps);
  for (int i = 0; i < num_qps; i++)
    for (isize j = 0; j < dim; ++j)
      CHECK(qps[usize(i)].results.x[j] == qps_vector[i].results.x[j]);

  // --- timings-parallel.cpp:151-176: `qps.push_back(qp)` COPIES an init-ed QP (value semantics): the
  // copy owns its own device state, the vector is solved in one launch per pool, and solving or
  // updating the copy leaves the original alone
  {
    std::vector<dense::QP<T>> copies;
    dense::QP<T> original{ dim, n_eq, n_in };
    original.settings.eps_abs = eps_abs;
    original.settings.eps_rel = 0;
    original.settings.initial_guess = InitialGuessStatus::NO_INITIAL_GUESS;
    const auto& m0 = models[0];
    original.init(m0.H, m0.g, m0.A, m0.b, m0.C, m0.l, m0.u);
    for (int i = 0; i < num_qps; i++) {
      dense::QP<T> qp{ dim, n_eq, n_in };
      qp.settings.eps_abs = eps_abs;
      qp.settings.eps_rel = 0;
      qp.settings.initial_guess = InitialGuessStatus::NO_INITIAL_GUESS;
      const auto& m = models[usize(i)];
      qp.init(m.H, m.g, m.A, m.b, m.C, m.l, m.u);
      copies.push_back(qp); // copy, then `qp` dies and gives its slot back
    }
    copies.push_back(original);
    CHECK(copies.back().pool().get() != nullptr);
    CHECK(!(copies.back().pool().get() == original.pool().get() && copies.back().slot() == original.slot()));
    dense::solve_in_parallel(copies);
    for (int i = 0; i < num_qps; i++)
      for (isize j = 0; j < dim; ++j)
        CHECK(copies[usize(i)].results.x[j] == qps_vector[i].results.x[j]);
    // the original was not solved by solving its copy ...
    CHECK(original.results.info.status == QPSolverOutput::PROXQP_MAX_ITER_REACHED);
    original.solve();
    // ... and gives the same answer as the copy (and as QP 0 of the batch) when it is
    for (isize j = 0; j < dim; ++j) {
      CHECK(original.results.x[j] == copies.back().results.x[j]);
      CHECK(original.results.x[j] == qps_vector[0].results.x[j]);
    }
    // copy assignment of a solved QP carries the solution (warm re-solve needs 0 iterations)
    dense::QP<T> assigned{ dim, n_eq, n_in };
    assigned = original;
    assigned.settings.initial_guess = InitialGuessStatus::WARM_START_WITH_PREVIOUS_RESULT;
    assigned.solve();
    CHECK(assigned.results.info.iter == 0);
  }

  // --- dense_qp_wrapper.cpp style: update g, warm start with the previous result, then from (x,y,z)
  {
    dense::QP<T>& qp = qps[0];
    dense::Model<T> m = models[0];
    for (isize j = 0; j < dim; ++j)
      m.g[j] *= 1.5;
    qp.settings.initial_guess = InitialGuessStatus::WARM_START_WITH_PREVIOUS_RESULT;
    qp.update(nullopt, m.g, nullopt, nullopt, nullopt, nullopt, nullopt);
    qp.solve();
    T pri, dua;
    residuals(m, qp.results, pri, dua);
    CHECK(pri <= eps_abs && dua <= eps_abs);
    CHECK(qp.model.g[0] == m.g[0]);
    Results<T> prev = qp.results;
    qp.solve(prev.x, prev.y, prev.z);
    CHECK(qp.settings.initial_guess == InitialGuessStatus::WARM_START); // helpers.hpp:727
    CHECK(qp.results.info.iter <= 1);
    // proximal parameters given at init (wrapper.hpp:354)
    qp.init(m.H, m.g, m.A, m.b, m.C, m.l, m.u, true, T(1e-7), T(1e-4), T(1e-2));
    CHECK(qp.results.info.rho == 1e-7 && qp.results.info.mu_eq == 1e-4 && qp.results.info.mu_in == 1e-2);
    qp.cleanup();
    CHECK(qp.results.x[0] == 0);
  }

  // --- dense_qp_wrapper.cpp:5052-5242: an update runs under the settings of the moment it is called.  Under the
  // default option it resets the results (helpers.hpp:522-531), so switching to WARM_START_WITH_PREVIOUS_RESULT
  // afterwards does not bring the old solution back: the next solve iterates again
  {
    dense::Model<T> m = models[1];
    dense::QP<T> qp(dim, n_eq, n_in);
    qp.settings.eps_abs = eps_abs;
    qp.settings.eps_rel = 0;
    qp.init(m.H, m.g, m.A, m.b, m.C, m.l, m.u, true, T(1e-7));
    qp.solve();
    CHECK(qp.results.info.iter > 0);
    qp.update(nullopt, nullopt, nullopt, nullopt, nullopt, nullopt, nullopt, true, T(1e-6));
    qp.settings.initial_guess = InitialGuessStatus::WARM_START_WITH_PREVIOUS_RESULT;
    CHECK(std::fabs(qp.settings.default_rho - 1e-6) <= 1e-9 && std::fabs(qp.results.info.rho - 1e-6) <= 1e-9);
    qp.solve();
    CHECK(qp.results.info.iter > 0);
    T pri, dua;
    residuals(m, qp.results, pri, dua);
    CHECK(pri <= eps_abs && dua <= eps_abs);
    qp.solve(); // and now the previous result IS the solution
    CHECK(qp.results.info.iter == 0);
  }

  // --- one-shot solve (dense_qp_solve.cpp)
  {
    const auto& m = models[1];
    Results<T> r = dense::solve<T>(m.H, m.g, m.A, m.b, m.C, m.l, m.u, nullopt, nullopt, nullopt, eps_abs, T(0));
    T pri, dua;
    residuals(m, r, pri, dua);
    CHECK(r.info.status == QPSolverOutput::PROXQP_SOLVED);
    CHECK(pri <= eps_abs && dua <= eps_abs);
  }

  // --- box constraints: z = [z_in; z_box] (dense_qp_wrapper.cpp:6889-6900)
  {
    dense::QP<T> qp(dim, 0, n_in, true);
    CHECK(qp.is_box_constrained());
    const auto& m = models[2];
    dense::Vec<T> lb(dim, -0.5), ub(dim, 0.5);
    qp.settings.eps_abs = eps_abs;
    qp.init(m.H, m.g, nullopt, nullopt, m.C, m.l, m.u, lb, ub);
    qp.solve();
    CHECK(qp.results.info.status == QPSolverOutput::PROXQP_SOLVED);
    CHECK(qp.results.z.size() == n_in + dim);
    for (isize j = 0; j < dim; ++j)
      CHECK(qp.results.x[j] <= 0.5 + 1e-9 && qp.results.x[j] >= -0.5 - 1e-9);
    bool threw = false;
    try {
      qp.init(m.H, m.g, nullopt, nullopt, m.C, m.l, m.u); // box QP without boxes (wrapper.hpp:367-372)
    } catch (const std::invalid_argument&) {
      threw = true;
    }
    CHECK(threw);
  }

  // --- argument errors are std::invalid_argument (wrapper.hpp:380-451, model.hpp:65-68)
  {
    bool threw = false;
    try {
      dense::QP<T> bad(0, 0, 0);
    } catch (const std::invalid_argument&) {
      threw = true;
    }
    CHECK(threw);
    threw = false;
    dense::QP<T> qp(4, 1, 2);
    dense::Mat<T> H3(3, 3);
    dense::Vec<T> g4(4);
    try {
      qp.init(H3, g4, nullopt, nullopt, nullopt, nullopt, nullopt);
    } catch (const std::invalid_argument&) {
      threw = true;
    }
    CHECK(threw);
  }

  // --- reference test/src/dense_qp_wrapper.cpp:7569-7591: a Hessian symmetric up to one ulp in one entry is a valid
  //     model (model.hpp:121-132: isApprox to machine precision), a plainly asymmetric one is std::invalid_argument
  {
    dense::Mat<T> S(3, 3);
    const T vals[3][3] = { { 0.4, -0.7, 0.2 }, { -0.7, 1.1, 0.5 }, { 0.2, 0.5, -0.3 } };
    for (isize i = 0; i < 3; ++i)
      for (isize j = 0; j < 3; ++j)
        S(i, j) = vals[i][j];
    S(0, 1) = S(1, 0) + std::numeric_limits<T>::epsilon();
    CHECK(S(0, 1) != S(1, 0));
    dense::QP<T> qp(3, 0, 0);
    qp.init(S, nullopt, nullopt, nullopt, nullopt, nullopt, nullopt);
    CHECK(qp.model.is_valid(false));
    qp.model.H(0, 2) += 1.0;
    bool threw = false;
    try {
      qp.model.is_valid(false);
    } catch (const std::invalid_argument&) {
      threw = true;
    }
    CHECK(threw);
    dense::QP<T> q5(3, 0, 2);
    threw = false;
    try {
      q5.model.is_valid(false); // C is zero, while n_in != 0 (model.hpp:144-145)
    } catch (const std::invalid_argument&) {
      threw = true;
    }
    CHECK(threw);
  }

  // --- strided (column-major) input is repacked: H^T of a symmetric H is H
  {
    const auto& m = models[3];
    dense::QP<T> qp(dim, n_eq, n_in);
    qp.settings.eps_abs = eps_abs;
    dense::MatRef<T> Hcm(m.H.data(), dim, dim, 1, dim);
    qp.init(Hcm, m.g, m.A, m.b, m.C, m.l, m.u);
    qp.solve();
    T pri, dua;
    residuals(m, qp.results, pri, dua);
    CHECK(pri <= eps_abs && dua <= eps_abs);
  }

  // --- dense_backward.cpp:16-80: dx/dg from compute_backward against central finite differences
  {
    const isize bd = 10, be = 5, bi = 0;
    utils::rand::set_seed(1);
    dense::Model<T> m = utils::dense_strongly_convex_qp(bd, be, bi, 0.85, 1e-1);
    dense::QP<T> qp{ bd, be, bi };
    qp.settings.eps_abs = eps_abs;
    qp.settings.eps_rel = 0;
    qp.init(m.H, m.g, m.A, m.b, nullopt, nullopt, nullopt);
    qp.solve();
    dense::Vec<T> loss_derivative(bd + be + bi);
    dense::Mat<T> dx_dg(bd, bd);
    for (isize i = 0; i < bd; ++i) {
      loss_derivative[i] = 1;
      dense::compute_backward<T>(qp, loss_derivative, 1e-5, 1e-7, 1e-7);
      for (isize j = 0; j < bd; ++j)
        dx_dg(i, j) = qp.model.backward_data.dL_dg[j];
      loss_derivative[i] = 0;
    }
    const T h = 1e-5;
    for (isize i = 0; i < bd; ++i) {
      dense::Vec<T> gp = m.g, gm = m.g;
      gp[i] += h;
      gm[i] -= h;
      dense::QP<T> qp2{ bd, be, bi };
      qp2.settings.eps_abs = eps_abs;
      qp2.init(m.H, gp, m.A, m.b, nullopt, nullopt, nullopt);
      qp2.solve();
      dense::Vec<T> xp = qp2.results.x;
      qp2.init(m.H, gm, m.A, m.b, nullopt, nullopt, nullopt);
      qp2.solve();
      for (isize r = 0; r < bd; ++r)
        CHECK(std::fabs((xp[r] - qp2.results.x[r]) / (2 * h) - dx_dg(r, i)) < 1e-5);
    }
    // batch form
    dense::BatchQP<T> bq(2);
    std::vector<dense::Vec<T>> lds;
    for (int k = 0; k < 2; ++k) {
      auto& q = bq.init_qp_in_place(bd, be, bi);
      q.settings.eps_abs = eps_abs;
      q.init(m.H, m.g, m.A, m.b, nullopt, nullopt, nullopt);
      dense::Vec<T> ld(bd + be + bi);
      ld[k] = 1;
      lds.push_back(ld);
    }
    dense::solve_in_parallel(bq);
    dense::qp_solve_backward_in_parallel<T>(nullopt, bq, lds, 1e-5, 1e-7, 1e-7);
    for (int k = 0; k < 2; ++k)
      for (isize j = 0; j < bd; ++j)
        CHECK(std::fabs(bq[k].model.backward_data.dL_dg[j] - dx_dg(k, j)) < 1e-9);
  }

  // --- Info timings (wrapper.hpp:374-377, 495-497; solver.hpp:1112-1115, 1783-1787): microseconds, non-zero and
  // ordered when settings.compute_timings is set, zero otherwise; a dirty re-solve clears setup_time
  {
    const auto& m = models[1];
    dense::QP<T> qp(dim, n_eq, n_in);
    qp.settings.eps_abs = eps_abs;
    qp.init(m.H, m.g, m.A, m.b, m.C, m.l, m.u);
    qp.solve();
    CHECK(qp.results.info.setup_time == 0 && qp.results.info.solve_time == 0 && qp.results.info.run_time == 0);
    dense::QP<T> qt(dim, n_eq, n_in);
    qt.settings.eps_abs = eps_abs;
    qt.settings.compute_timings = true;
    qt.init(m.H, m.g, m.A, m.b, m.C, m.l, m.u);
    CHECK(qt.results.info.setup_time > 0);
    qt.solve();
    const auto& i = qt.results.info;
    CHECK(i.setup_time > 0 && i.solve_time > 0 && i.solve_time < 60e6);
    CHECK(std::fabs(i.run_time - (i.setup_time + i.solve_time)) <= 1e-9 * i.run_time);
    qt.solve();
    CHECK(qt.results.info.setup_time == 0 && qt.results.info.solve_time > 0 &&
          qt.results.info.run_time == qt.results.info.solve_time);
  }

  // --- a NEW QP on a recycled registry slot starts from defaults (reference wrapper.hpp:140-333): neither the
  // settings nor the model of the QP that owned the slot before may show through
  {
    const auto& m = models[0];
    {
      dense::QP<T> a(dim, n_eq, n_in);
      a.settings.eps_abs = 1e-3;
      a.settings.max_iter = 7;
      a.settings.default_rho = 1e-2;
      a.settings.initial_guess = InitialGuessStatus::WARM_START_WITH_PREVIOUS_RESULT;
      a.init(m.H, m.g, m.A, m.b, m.C, m.l, m.u);
      a.solve();
    } // a dies: its slot goes back to the pool
    dense::QP<T> fresh(dim, n_eq, n_in); // a slot nobody used
    dense::QP<T> b(dim, n_eq, n_in);     // (one of the two sits on a's slot, whatever the pool layout)
    for (dense::QP<T>* q : { &fresh, &b }) {
      CHECK(q->settings.eps_abs == 1e-5);
      CHECK(q->settings.max_iter == 10000);
      CHECK(q->settings.default_rho == 1e-6);
      CHECK(q->settings.initial_guess == InitialGuessStatus::EQUALITY_CONSTRAINED_INITIAL_GUESS);
      CHECK(q->results.info.rho == 1e-6);
      CHECK(q->results.info.mu_eq == 1e-3);
      CHECK(q->results.info.status == QPSolverOutput::PROXQP_NOT_RUN);
    }
    // stale model arrays must not survive either: init with H and g only -> the unconstrained minimiser
    // of THIS model, not a's constraints
    dense::QP<T> c(dim, n_eq, n_in);
    c.settings.eps_abs = eps_abs;
    c.init(m.H, m.g, nullopt, nullopt, nullopt, nullopt, nullopt);
    c.solve();
    T worst = 0;
    for (isize j = 0; j < dim; ++j) {
      T r = m.g[j];
      for (isize k = 0; k < dim; ++k)
        r += m.H(j, k) * c.results.x[k];
      worst = std::max(worst, std::fabs(r));
    }
    CHECK(worst <= 1e-7); // H x + g = 0: no stale A, C, l, u took part
  }

  // --- independent QP objects driven from several host threads (legal with the reference:
  // `#pragma omp parallel for` over qps[i].solve()): same answers as the serial run
  {
    std::vector<dense::QP<T>> qs;
    for (usize i = 0; i < 8; ++i) {
      qs.emplace_back(dim, n_eq, n_in);
      qs.back().settings.eps_abs = eps_abs;
    }
    std::vector<std::thread> th;
    for (usize t = 0; t < 4; ++t)
      th.emplace_back([&, t]() {
        for (usize i = t; i < 8; i += 4) {
          const auto& m = models[i % models.size()];
          qs[i].init(m.H, m.g, m.A, m.b, m.C, m.l, m.u);
          qs[i].solve();
        }
      });
    for (auto& t : th)
      t.join();
    for (usize i = 0; i < 8; ++i) {
      const auto& m = models[i % models.size()];
      dense::QP<T> ref(dim, n_eq, n_in);
      ref.settings.eps_abs = eps_abs;
      ref.init(m.H, m.g, m.A, m.b, m.C, m.l, m.u);
      ref.solve();
      for (isize j = 0; j < dim; ++j)
        CHECK(qs[i].results.x[j] == ref.results.x[j]);
      CHECK(qs[i].results.info.status == QPSolverOutput::PROXQP_SOLVED);
    }
  }

  // --- a BatchQP spread over several devices from ONE process (pqp_multi_*; here three logical shards on device 0,
  // the way the path is exercised on a one-GPU box): same answers bit for bit as the one-device BatchQP above, QPs
  // land on the shards in contiguous ranges, and two signatures are in flight together
  {
    dense::BatchQP<T> multi(usize(num_qps + 1), std::vector<int>{ 0, 0, 0 });
    CHECK(multi.devices().size() == 3);
    for (int i = 0; i < num_qps; i++) {
      const auto& m = models[usize(i)];
      auto& qp = multi.init_qp_in_place(dim, n_eq, n_in);
      qp.settings.eps_abs = eps_abs;
      qp.settings.eps_rel = 0;
      qp.settings.initial_guess = InitialGuessStatus::NO_INITIAL_GUESS;
      qp.init(m.H, m.g, m.A, m.b, m.C, m.l, m.u);
    }
    // a second signature in the same container: its own multi-device batch, launched alongside
    utils::rand::set_seed(77);
    dense::Model<T> other = utils::dense_strongly_convex_qp(dim + 2, n_eq, n_in + 1, sparsity_factor, strong_convexity_factor);
    auto& qo = multi.init_qp_in_place(dim + 2, n_eq, n_in + 1);
    qo.settings.eps_abs = eps_abs;
    qo.settings.eps_rel = 0;
    qo.init(other.H, other.g, other.A, other.b, other.C, other.l, other.u);
    CHECK(multi.pools().size() >= 4); // three shards of the first signature + the (one-QP) second
    usize placed = 0;
    for (const auto& e : multi.pools())
      placed += e.members.size();
    CHECK(isize(placed) == multi.size());
    dense::solve_in_parallel(multi);
    for (int i = 0; i < num_qps; i++) {
      CHECK(multi[i].results.info.status == QPSolverOutput::PROXQP_SOLVED);
      CHECK(multi[i].results.info.iter == qps_vector[i].results.info.iter);
      for (isize j = 0; j < dim; ++j)
        CHECK(multi[i].results.x[j] == qps_vector[i].results.x[j]);
      for (isize j = 0; j < n_in; ++j)
        CHECK(multi[i].results.z[j] == qps_vector[i].results.z[j]);
    }
    {
      T pri, dua;
      residuals(other, qo.results, pri, dua);
      CHECK(qo.results.info.status == QPSolverOutput::PROXQP_SOLVED);
      CHECK(pri <= eps_abs && dua <= eps_abs);
    }
    // per-QP methods address the right shard: update one QP of the last shard, re-solve it alone
    auto& last = multi[num_qps - 1];
    dense::Vec<T> g2 = models[usize(num_qps - 1)].g;
    for (isize j = 0; j < dim; ++j)
      g2[j] *= 2;
    last.update(nullopt, g2, nullopt, nullopt, nullopt, nullopt, nullopt);
    last.solve();
    dense::QP<T> ref(dim, n_eq, n_in);
    ref.settings.eps_abs = eps_abs;
    ref.settings.eps_rel = 0;
    ref.settings.initial_guess = InitialGuessStatus::NO_INITIAL_GUESS;
    {
      const auto& m = models[usize(num_qps - 1)];
      ref.init(m.H, m.g, m.A, m.b, m.C, m.l, m.u);
      ref.solve();
      ref.update(nullopt, g2, nullopt, nullopt, nullopt, nullopt, nullopt);
      ref.solve();
    }
    for (isize j = 0; j < dim; ++j)
      CHECK(last.results.x[j] == ref.results.x[j]);
    // default constructor: every visible device (one here unless the box has more)
    dense::BatchQP<T> dflt(4);
    CHECK(dflt.devices().empty() ? dflt.device() == 0 : dflt.devices().size() > 1);
  }

  // --- solve_backward_in_parallel over a std::vector of QPs (reference parallel/qp_solve.hpp:83-110): one launch per
  // pool (pqp_batch_backward_subset), same numbers as compute_backward QP by QP
  {
    std::vector<dense::QP<T>> vq, vr;
    std::vector<dense::Vec<T>> lds;
    for (int k = 0; k < 3; ++k) {
      const auto& m = models[usize(k)];
      for (auto* v : { &vq, &vr }) {
        v->emplace_back(dim, n_eq, n_in);
        v->back().settings.eps_abs = eps_abs;
        v->back().init(m.H, m.g, m.A, m.b, m.C, m.l, m.u);
      }
      dense::Vec<T> ld(dim + n_eq + n_in);
      ld[k] = 1;
      ld[dim + 1] = 0.5;
      lds.push_back(ld);
    }
    dense::solve_in_parallel(vq);
    dense::solve_in_parallel(vr);
    dense::qp_solve_backward_in_parallel<T>(nullopt, vq, lds, 1e-5, 1e-7, 1e-7);
    for (usize k = 0; k < 3; ++k) {
      dense::compute_backward<T>(vr[k], lds[k], 1e-5, 1e-7, 1e-7);
      for (isize j = 0; j < dim; ++j)
        CHECK(vq[k].model.backward_data.dL_dg[j] == vr[k].model.backward_data.dL_dg[j]);
      for (isize j = 0; j < n_in; ++j)
        CHECK(vq[k].model.backward_data.dL_du[j] == vr[k].model.backward_data.dL_du[j]);
    }
  }

  std::printf("facade_test: %d failure(s)\n", failures);
  return failures == 0 ? 0 : 1;
}
