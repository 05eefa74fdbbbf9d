// TEST INFRASTRUCTURE ONLY -- see proxqp_oracle.hpp.  CPU restatement of the
// reference dense ProxQP solver (PrimalDualLDLT backend); every function cites
// the reference lines it follows (paths relative to
// /root/reference/include/proxsuite/proxqp/).
#include "proxqp_oracle.hpp"

#include <chrono>
#include <cstdio>
#include <cstdlib>

namespace pqo {

namespace {

constexpr double MACHINE_EPS = std::numeric_limits<double>::epsilon();

inline bool
is_absent(double v)
{
  return std::isnan(v);
}

// ---------------------------------------------------------------- results.hpp
// results.hpp:157-174
void
cleanup_statistics(pqp_info& info)
{
  info.run_time = 0;
  info.setup_time = 0;
  info.solve_time = 0;
  info.objValue = 0.;
  info.iter = 0;
  info.iter_ext = 0;
  info.mu_updates = 0;
  info.rho_updates = 0;
  info.pri_res = 0.;
  info.dua_res = 0.;
  info.duality_gap = 0.;
  info.iterative_residual = 0.;
  info.status = PQP_MAX_ITER_REACHED;
}
// results.hpp:175-194 (always called with settings in the dense path)
void
cold_start(pqp_info& info, const pqp_settings& s)
{
  info.nu = 1.;
  info.rho = s.default_rho;
  info.mu_eq = s.default_mu_eq;
  info.mu_eq_inv = 1.0 / info.mu_eq;
  info.mu_in = s.default_mu_in;
  info.mu_in_inv = 1.0 / info.mu_in;
  info.minimal_H_eigenvalue_estimate = s.default_H_eigenvalue_estimate;
  cleanup_statistics(info);
}
void
zero(Vec& v)
{
  std::fill(v.begin(), v.end(), 0.0);
}
// results.hpp:148-156
void
results_cleanup(Results& r, const pqp_settings& s)
{
  zero(r.x);
  zero(r.y);
  zero(r.z);
  zero(r.se);
  zero(r.si);
  cold_start(r.info, s);
}
// results.hpp:195-203
void
results_cleanup_all_except_prox_parameters(Results& r)
{
  zero(r.x);
  zero(r.y);
  zero(r.z);
  zero(r.se);
  zero(r.si);
  cleanup_statistics(r.info);
}

// ------------------------------------------------------------- workspace.hpp
// workspace.hpp:330-377
void
work_cleanup(Workspace& w, isize n_constraints)
{
  zero(w.H_scaled);
  zero(w.g_scaled);
  zero(w.A_scaled);
  zero(w.C_scaled);
  zero(w.b_scaled);
  zero(w.u_scaled);
  zero(w.l_scaled);
  zero(w.Hdx);
  zero(w.Cdx);
  zero(w.Adx);
  zero(w.active_part_z);
  zero(w.dw_aug);
  zero(w.rhs);
  zero(w.err);
  w.alpha = 1.;
  zero(w.dual_residual_scaled);
  zero(w.primal_residual_in_scaled_up);
  zero(w.primal_residual_in_scaled_up_plus_alphaCdx);
  zero(w.primal_residual_in_scaled_low_plus_alphaCdx);
  zero(w.CTz);
  zero(w.x_prev);
  zero(w.y_prev);
  zero(w.z_prev);
  for (isize i = 0; i < n_constraints; i++) {
    w.current_bijection_map[size_t(i)] = i;
    w.new_bijection_map[size_t(i)] = i;
    w.active_inequalities[size_t(i)] = 0;
  }
  w.constraints_changed = false;
  w.dirty = false;
  w.refactorize = false;
  w.proximal_parameter_update = false;
  w.is_initialized = false;
  w.n_c = 0;
}

// ------------------------------------------------------ preconditioner/ruiz.hpp
// ruiz.hpp:29-311 (Symmetry::general, the only one the dense QP uses: :346)
double
ruiz_scale_qp_in_place(QP& qp, double epsilon, isize max_iter)
{
  Workspace& w = qp.work;
  const isize n = qp.model.dim, n_eq = qp.model.n_eq, n_in = qp.model.n_in;
  const bool box = qp.box_constraints;
  const bool infeas = qp.settings.primal_infeasibility_solving != 0;
  const isize n_constraints = qp.n_constraints();
  double* H = w.H_scaled.data();
  double* A = w.A_scaled.data();
  double* C = w.C_scaled.data();
  double c = 1;
  Vec& S = qp.ruiz.delta;
  if (box)
    std::fill(w.i_scaled.begin(), w.i_scaled.end(), 1.0);
  double gamma = 1;
  Vec delta(size_t(n + n_eq + n_constraints), 0.0); // LDLT_TEMP_VEC is zero-initialised
  isize iter = 1;
  auto err = [&]() {
    double r = 0;
    for (double d : delta)
      r = std::max(r, std::fabs(1 - d));
    return r;
  };
  Vec colA(static_cast<size_t>(n)), colC(static_cast<size_t>(n)), colH(static_cast<size_t>(n));
  while (err() > epsilon) {
    if (iter == max_iter)
      break;
    ++iter;
    // column norms (ruiz.hpp:94-169)
    std::fill(colA.begin(), colA.end(), 0.0);
    std::fill(colC.begin(), colC.end(), 0.0);
    std::fill(colH.begin(), colH.end(), 0.0);
    for (isize i = 0; i < n_eq; ++i)
      for (isize k = 0; k < n; ++k)
        colA[size_t(k)] = std::max(colA[size_t(k)], std::fabs(A[i * n + k]));
    for (isize i = 0; i < n_in; ++i)
      for (isize k = 0; k < n; ++k)
        colC[size_t(k)] = std::max(colC[size_t(k)], std::fabs(C[i * n + k]));
    if (qp.hessian_type == PQP_HESSIAN_DENSE) {
      for (isize i = 0; i < n; ++i)
        for (isize k = 0; k < n; ++k)
          colH[size_t(k)] = std::max(colH[size_t(k)], std::fabs(H[i * n + k]));
    } else if (qp.hessian_type == PQP_HESSIAN_DIAGONAL) {
      for (isize k = 0; k < n; ++k)
        colH[size_t(k)] = std::fabs(H[k * n + k]);
    }
    for (isize k = 0; k < n; ++k) {
      double m = 0;
      if (qp.hessian_type != PQP_HESSIAN_ZERO)
        m = colH[size_t(k)];
      m = std::max(m, n_eq > 0 ? colA[size_t(k)] : 0.0);
      m = std::max(m, n_in > 0 ? colC[size_t(k)] : 0.0);
      m = std::max(m, box ? w.i_scaled[size_t(k)] : 0.0);
      double aux = std::sqrt(m);
      delta[size_t(k)] = (aux == 0.0) ? 1.0 : 1.0 / (aux + MACHINE_EPS);
    }
    // row norms (ruiz.hpp:170-194)
    if (infeas) {
      for (isize k = n; k < n + n_eq + n_constraints; ++k)
        delta[size_t(k)] = 1.0;
    } else {
      for (isize k = 0; k < n_eq; ++k) {
        double aux = std::sqrt(infty_norm(A + k * n, n));
        delta[size_t(n + k)] = (aux == 0.0) ? 1.0 : 1.0 / (aux + MACHINE_EPS);
      }
      for (isize k = 0; k < n_in; ++k) {
        double aux = std::sqrt(infty_norm(C + k * n, n));
        delta[size_t(k + n + n_eq)] = (aux == 0.0) ? 1.0 : 1.0 / (aux + MACHINE_EPS);
      }
      if (box)
        for (isize k = 0; k < n; ++k)
          delta[size_t(k + n + n_eq + n_in)] = 1.0 / std::sqrt(w.i_scaled[size_t(k)] + MACHINE_EPS);
    }
    // apply (ruiz.hpp:202-308)
    for (isize i = 0; i < n_eq; ++i)
      for (isize k = 0; k < n; ++k)
        A[i * n + k] = delta[size_t(n + i)] * A[i * n + k] * delta[size_t(k)];
    for (isize i = 0; i < n_in; ++i)
      for (isize k = 0; k < n; ++k)
        C[i * n + k] = delta[size_t(n + n_eq + i)] * C[i * n + k] * delta[size_t(k)];
    if (box) {
      const double* dtail = delta.data() + n + n_eq + n_in;
      for (isize k = 0; k < n; ++k) {
        w.i_scaled[size_t(k)] *= delta[size_t(k)];
        w.i_scaled[size_t(k)] *= dtail[k];
        w.u_box_scaled[size_t(k)] *= dtail[k];
        w.l_box_scaled[size_t(k)] *= dtail[k];
      }
    }
    for (isize k = 0; k < n; ++k)
      w.g_scaled[size_t(k)] *= delta[size_t(k)];
    for (isize k = 0; k < n_eq; ++k)
      w.b_scaled[size_t(k)] *= delta[size_t(n + k)];
    for (isize k = 0; k < n_in; ++k) {
      w.u_scaled[size_t(k)] *= delta[size_t(n + n_eq + k)];
      w.l_scaled[size_t(k)] *= delta[size_t(n + n_eq + k)];
    }
    switch (qp.hessian_type) {
      case PQP_HESSIAN_ZERO:
        break;
      case PQP_HESSIAN_DENSE: {
        for (isize i = 0; i < n; ++i)
          for (isize k = 0; k < n; ++k)
            H[i * n + k] = delta[size_t(i)] * H[i * n + k] * delta[size_t(k)];
        std::fill(colH.begin(), colH.end(), 0.0);
        for (isize i = 0; i < n; ++i)
          for (isize k = 0; k < n; ++k)
            colH[size_t(k)] = std::max(colH[size_t(k)], std::fabs(H[i * n + k]));
        double mean = 0;
        for (isize k = 0; k < n; ++k)
          mean += colH[size_t(k)];
        mean /= double(n);
        gamma = 1 / std::max(1.0, mean);
        // NB: gamma is NOT applied to a Dense H by the reference (ruiz.hpp:256-287 computes it and breaks;
        // only the Diagonal case has `H *= gamma`, :301), although g (:304) and c (:307) take it and the
        // non-executing path multiplies H by c (:505).  Restated as written.
        break;
      }
      case PQP_HESSIAN_DIAGONAL: {
        double mx = 0;
        for (isize k = 0; k < n; ++k) {
          H[k * n + k] *= delta[size_t(k)];
          H[k * n + k] *= delta[size_t(k)];
          mx = std::max(mx, std::fabs(H[k * n + k]));
        }
        gamma = 1 / std::max(1.0, mx / double(n));
        for (isize k = 0; k < n * n; ++k)
          H[k] *= gamma;
        break;
      }
    }
    for (isize k = 0; k < n; ++k)
      w.g_scaled[size_t(k)] *= gamma;
    for (size_t k = 0; k < delta.size(); ++k)
      S[k] *= delta[k];
    c *= gamma;
    qp.counters.level2_flops += 3.0 * double(n * n + n_eq * n + n_in * n);
  }
  return c;
}

// ruiz.hpp:403-512
void
ruiz_scale_qp(QP& qp, bool execute_preconditioner)
{
  Workspace& w = qp.work;
  Ruiz& rz = qp.ruiz;
  const isize n = qp.model.dim, n_eq = qp.model.n_eq, n_in = qp.model.n_in;
  if (execute_preconditioner) {
    std::fill(rz.delta.begin(), rz.delta.end(), 1.0);
    rz.c = ruiz_scale_qp_in_place(
      qp, qp.settings.preconditioner_accuracy, qp.settings.preconditioner_max_iter);
  } else {
    const Vec& d = rz.delta;
    // DELIBERATE DEVIATION (see copy_model_to_scaled): i_scaled restarts at 1
    // so that re-applying the stored scaling is idempotent.
    if (qp.box_constraints)
      std::fill(w.i_scaled.begin(), w.i_scaled.end(), 1.0);
    double* H = w.H_scaled.data();
    double* A = w.A_scaled.data();
    double* C = w.C_scaled.data();
    for (isize i = 0; i < n_eq; ++i)
      for (isize k = 0; k < n; ++k)
        A[i * n + k] = d[size_t(n + i)] * A[i * n + k] * d[size_t(k)];
    for (isize i = 0; i < n_in; ++i)
      for (isize k = 0; k < n; ++k)
        C[i * n + k] = d[size_t(n + n_eq + i)] * C[i * n + k] * d[size_t(k)];
    switch (qp.hessian_type) {
      case PQP_HESSIAN_DENSE:
        for (isize i = 0; i < n; ++i)
          for (isize k = 0; k < n; ++k)
            H[i * n + k] = d[size_t(i)] * H[i * n + k] * d[size_t(k)];
        break;
      case PQP_HESSIAN_ZERO:
        break;
      case PQP_HESSIAN_DIAGONAL:
        for (isize k = 0; k < n; ++k) {
          H[k * n + k] *= d[size_t(k)];
          H[k * n + k] *= d[size_t(k)];
        }
        break;
    }
    for (isize k = 0; k < n; ++k)
      w.g_scaled[size_t(k)] *= d[size_t(k)];
    for (isize k = 0; k < n_eq; ++k)
      w.b_scaled[size_t(k)] *= d[size_t(n + k)];
    for (isize k = 0; k < n_in; ++k) {
      w.l_scaled[size_t(k)] *= d[size_t(n + n_eq + k)];
      w.u_scaled[size_t(k)] *= d[size_t(n + n_eq + k)];
    }
    if (qp.box_constraints) {
      const double* dtail = d.data() + n + n_eq + n_in;
      for (isize k = 0; k < n; ++k) {
        w.u_box_scaled[size_t(k)] *= dtail[k];
        w.l_box_scaled[size_t(k)] *= dtail[k];
        w.i_scaled[size_t(k)] *= dtail[k];
        w.i_scaled[size_t(k)] *= d[size_t(k)];
      }
    }
    for (isize k = 0; k < n; ++k)
      w.g_scaled[size_t(k)] *= rz.c;
    for (isize k = 0; k < n * n; ++k)
      H[k] *= rz.c;
  }
}

// Ruiz (un)scaling helpers, ruiz.hpp:518-694
struct Scaler
{
  const Ruiz& r;
  isize n, n_eq, n_in;
  explicit Scaler(const QP& qp)
    : r(qp.ruiz)
    , n(qp.model.dim)
    , n_eq(qp.model.n_eq)
    , n_in(qp.model.n_in)
  {
  }
  const double* dx() const { return r.delta.data(); }
  const double* deq() const { return r.delta.data() + n; }
  const double* din() const { return r.delta.data() + n + n_eq; }
  const double* dbox() const { return r.delta.data() + r.delta.size() - size_t(n); }
  void scale_primal(double* v) const
  {
    for (isize i = 0; i < n; ++i)
      v[i] /= dx()[i];
  }
  void unscale_primal(double* v) const
  {
    for (isize i = 0; i < n; ++i)
      v[i] *= dx()[i];
  }
  void scale_dual_eq(double* v) const
  {
    for (isize i = 0; i < n_eq; ++i)
      v[i] = v[i] / deq()[i] * r.c;
  }
  void unscale_dual_eq(double* v) const
  {
    for (isize i = 0; i < n_eq; ++i)
      v[i] = v[i] * deq()[i] / r.c;
  }
  void scale_dual_in(double* v) const
  {
    for (isize i = 0; i < n_in; ++i)
      v[i] = v[i] / din()[i] * r.c;
  }
  void unscale_dual_in(double* v) const
  {
    for (isize i = 0; i < n_in; ++i)
      v[i] = v[i] * din()[i] / r.c;
  }
  void scale_box_dual_in(double* v) const
  {
    for (isize i = 0; i < n; ++i)
      v[i] = v[i] / dbox()[i] * r.c;
  }
  void unscale_box_dual_in(double* v) const
  {
    for (isize i = 0; i < n; ++i)
      v[i] = dbox()[i] * v[i] / r.c;
  }
  void scale_primal_residual_eq(double* v) const
  {
    for (isize i = 0; i < n_eq; ++i)
      v[i] *= deq()[i];
  }
  void unscale_primal_residual_eq(double* v) const
  {
    for (isize i = 0; i < n_eq; ++i)
      v[i] /= deq()[i];
  }
  void scale_primal_residual_in(double* v) const
  {
    for (isize i = 0; i < n_in; ++i)
      v[i] *= din()[i];
  }
  void unscale_primal_residual_in(double* v) const
  {
    for (isize i = 0; i < n_in; ++i)
      v[i] /= din()[i];
  }
  void scale_box_primal_residual_in(double* v) const
  {
    for (isize i = 0; i < n; ++i)
      v[i] *= dbox()[i];
  }
  void unscale_box_primal_residual_in(double* v) const
  {
    for (isize i = 0; i < n; ++i)
      v[i] /= dbox()[i];
  }
  void scale_dual_residual(double* v) const
  {
    for (isize i = 0; i < n; ++i)
      v[i] *= dx()[i] * r.c;
  }
  void unscale_dual_residual(double* v) const
  {
    for (isize i = 0; i < n; ++i)
      v[i] /= dx()[i] * r.c;
  }
};

// ----------------------------------------------------------------- helpers.hpp
// helpers.hpp:298-329
void
setup_equilibration(QP& qp, bool execute_preconditioner)
{
  ruiz_scale_qp(qp, execute_preconditioner);
  qp.work.correction_guess_rhs_g = infty_norm(qp.work.g_scaled.data(), qp.model.dim);
}

// copies model -> *_scaled with the 1e20 clamp, helpers.hpp:614-651
void
copy_model_to_scaled(QP& qp, bool clamp)
{
  Workspace& w = qp.work;
  Model& m = qp.model;
  if (qp.hessian_type != PQP_HESSIAN_ZERO)
    w.H_scaled = m.H;
  w.g_scaled = m.g;
  w.A_scaled = m.A;
  w.b_scaled = m.b;
  w.C_scaled = m.C;
  if (clamp) {
    for (isize i = 0; i < m.n_in; ++i) {
      w.u_scaled[size_t(i)] = (m.u[size_t(i)] <= 1.E20) ? m.u[size_t(i)] : 1.E20;
      w.l_scaled[size_t(i)] = (m.l[size_t(i)] >= -1.E20) ? m.l[size_t(i)] : -1.E20;
    }
    if (qp.box_constraints)
      for (isize i = 0; i < m.dim; ++i) {
        w.u_box_scaled[size_t(i)] = (m.u_box[size_t(i)] <= 1.E20) ? m.u_box[size_t(i)] : 1.E20;
        w.l_box_scaled[size_t(i)] = (m.l_box[size_t(i)] >= -1.E20) ? m.l_box[size_t(i)] : -1.E20;
      }
  } else {
    // solver.hpp:1192-1207 (dirty re-solve): plain copies, no clamp.
    w.u_scaled = m.u;
    w.l_scaled = m.l;
    // DELIBERATE DEVIATION: the reference does not re-copy the box bounds nor
    // reset i_scaled here, so a second solve() of a box-constrained QP
    // re-applies delta to already-scaled u_box/l_box/i_scaled (latent bug,
    // invisible to its tests because their boxes are inactive).  We re-copy.
    if (qp.box_constraints)
      for (isize i = 0; i < m.dim; ++i) {
        w.u_box_scaled[size_t(i)] = (m.u_box[size_t(i)] <= 1.E20) ? m.u_box[size_t(i)] : 1.E20;
        w.l_box_scaled[size_t(i)] = (m.l_box[size_t(i)] >= -1.E20) ? m.l_box[size_t(i)] : -1.E20;
      }
  }
}

// helpers.hpp:500-667
void
setup(QP& qp, const double* H, const double* g, const double* A, const double* b,
      const double* C, const double* l, const double* u, const double* l_box,
      const double* u_box, int preconditioner_status /*0 EXECUTE 1 KEEP 2 IDENTITY*/)
{
  Workspace& w = qp.work;
  Model& m = qp.model;
  Results& r = qp.results;
  const isize n = m.dim, n_eq = m.n_eq, n_in = m.n_in;
  const isize nc = qp.n_constraints();
  switch (qp.settings.initial_guess) {
    case PQP_EQUALITY_CONSTRAINED_INITIAL_GUESS:
    case PQP_NO_INITIAL_GUESS:
    case PQP_WARM_START:
      if (w.proximal_parameter_update)
        results_cleanup_all_except_prox_parameters(r);
      else
        results_cleanup(r, qp.settings);
      work_cleanup(w, nc);
      break;
    case PQP_COLD_START_WITH_PREVIOUS_RESULT:
      if (w.proximal_parameter_update)
        cleanup_statistics(r.info);
      else
        cold_start(r.info, qp.settings);
      work_cleanup(w, nc);
      break;
    case PQP_WARM_START_WITH_PREVIOUS_RESULT:
      if (w.refactorize || w.proximal_parameter_update) {
        work_cleanup(w, nc);
        w.refactorize = true;
      }
      cleanup_statistics(r.info);
      break;
  }
  if (H)
    m.H.assign(H, H + n * n);
  if (g)
    m.g.assign(g, g + n);
  if (A)
    m.A.assign(A, A + n_eq * n);
  if (b)
    m.b.assign(b, b + n_eq);
  if (C)
    m.C.assign(C, C + n_in * n);
  if (u)
    m.u.assign(u, u + n_in);
  if (l)
    m.l.assign(l, l + n_in);
  if (u_box)
    m.u_box.assign(u_box, u_box + n);
  if (l_box)
    m.l_box.assign(l_box, l_box + n);
  copy_model_to_scaled(qp, true);
  w.dual_feasibility_rhs_2 = infty_norm(m.g.data(), n);
  setup_equilibration(qp, preconditioner_status == 0);
}

// helpers.hpp:678-705
void
update_proximal_parameters(QP& qp, double rho_new, double mu_eq_new, double mu_in_new)
{
  if (!is_absent(rho_new)) {
    qp.settings.default_rho = rho_new;
    qp.results.info.rho = rho_new;
    qp.work.proximal_parameter_update = true;
  }
  if (!is_absent(mu_eq_new)) {
    qp.settings.default_mu_eq = mu_eq_new;
    qp.results.info.mu_eq = mu_eq_new;
    qp.results.info.mu_eq_inv = 1.0 / mu_eq_new;
    qp.work.proximal_parameter_update = true;
  }
  if (!is_absent(mu_in_new)) {
    qp.settings.default_mu_in = mu_in_new;
    qp.results.info.mu_in = mu_in_new;
    qp.results.info.mu_in_inv = 1.0 / mu_in_new;
    qp.work.proximal_parameter_update = true;
  }
}
// helpers.hpp:174-189
void
update_default_rho_with_minimal_Hessian_eigen_value(QP& qp, double manual)
{
  if (!is_absent(manual)) {
    qp.settings.default_H_eigenvalue_estimate = manual;
    qp.results.info.minimal_H_eigenvalue_estimate = manual;
  }
  qp.settings.default_rho += std::fabs(qp.results.info.minimal_H_eigenvalue_estimate);
  qp.results.info.rho = qp.settings.default_rho;
}

// helpers.hpp:239-285 (PrimalDualLDLT)
void
setup_factorization(QP& qp)
{
  Workspace& w = qp.work;
  const isize n = qp.model.dim, n_eq = qp.model.n_eq;
  const isize m = n + n_eq;
  double* kkt = w.kkt.data();
  for (isize i = 0; i < n; ++i)
    for (isize j = 0; j < n; ++j)
      kkt[i * m + j] = (qp.hessian_type == PQP_HESSIAN_ZERO) ? 0.0 : w.H_scaled[size_t(i * n + j)];
  for (isize i = 0; i < n; ++i)
    kkt[i * m + i] += qp.results.info.rho;
  for (isize i = 0; i < n_eq; ++i)
    for (isize j = 0; j < n; ++j) {
      kkt[j * m + n + i] = w.A_scaled[size_t(i * n + j)];
      kkt[(n + i) * m + j] = w.A_scaled[size_t(i * n + j)];
    }
  for (isize i = 0; i < n_eq; ++i)
    for (isize j = 0; j < n_eq; ++j)
      kkt[(n + i) * m + n + j] = 0;
  for (isize i = 0; i < n_eq; ++i)
    kkt[(n + i) * m + n + i] = -qp.results.info.mu_eq;
  // ldl.factorize(kkt.transpose()): lower triangle of kkt^T (col-major) ==
  // entries kkt(j,i) (row-major) for i>=j.
  w.ldl.factorize(m, [&](isize i, isize j) { return kkt[j * m + i]; });
}

// ------------------------------------------------------------------ solver.hpp
// solver.hpp:38-115 (PrimalDualLDLT)
void
refactorize(QP& qp, double rho_new)
{
  Workspace& w = qp.work;
  Results& r = qp.results;
  if (!w.constraints_changed && rho_new == r.info.rho)
    return;
  ++qp.n_refactorize;
  const isize n = qp.model.dim, n_eq = qp.model.n_eq, n_in = qp.model.n_in;
  const isize m = n + n_eq;
  const isize nc = qp.n_constraints();
  double* kkt = w.kkt.data();
  for (isize i = 0; i < n; ++i)
    kkt[i * m + i] += rho_new - r.info.rho;
  for (isize i = 0; i < n_eq; ++i)
    kkt[(n + i) * m + n + i] = -r.info.mu_eq;
  w.ldl.factorize(m, [&](isize i, isize j) { return kkt[j * m + i]; });
  isize n_c = w.n_c;
  isize rows = n + n_eq + n_c;
  Vec new_cols(size_t(rows * std::max<isize>(n_c, 1)), 0.0);
  for (isize i = 0; i < nc; ++i) {
    isize j = w.current_bijection_map[size_t(i)];
    if (j < n_c) {
      double* col = new_cols.data() + j * rows;
      if (i >= n_in) {
        col[i - n_in] = w.i_scaled[size_t(i - n_in)];
      } else {
        for (isize k = 0; k < n; ++k)
          col[k] = w.C_scaled[size_t(i * n + k)];
      }
      for (isize k = n; k < rows; ++k)
        col[k] = 0;
      col[n + n_eq + j] = -r.info.mu_in;
    }
  }
  w.ldl.insert_block_at(n + n_eq, new_cols.data(), rows, n_c);
  qp.n_inserted += n_c;
  w.constraints_changed = false;
}

// solver.hpp:128-232 (PrimalDualLDLT)
void
mu_update(QP& qp, double mu_eq_new, double mu_in_new)
{
  Workspace& w = qp.work;
  Results& r = qp.results;
  const isize n = qp.model.dim, n_eq = qp.model.n_eq;
  isize n_c = w.n_c;
  if ((n_eq + n_c) == 0)
    return;
  Vec rank_update_alpha(size_t(n_eq + n_c));
  for (isize k = 0; k < n_eq; ++k)
    rank_update_alpha[size_t(k)] = r.info.mu_eq - mu_eq_new;
  for (isize k = 0; k < n_c; ++k)
    rank_update_alpha[size_t(n_eq + k)] = r.info.mu_in - mu_in_new;
  std::vector<isize> indices(size_t(n_eq + n_c));
  for (isize k = 0; k < n_eq; ++k)
    indices[size_t(k)] = n + k;
  for (isize k = 0; k < n_c; ++k)
    indices[size_t(n_eq + k)] = n + n_eq + k;
  w.ldl.diagonal_update_clobber_indices(indices.data(), n_eq + n_c, rank_update_alpha.data());
  w.constraints_changed = true;
}

// y = sym(H_lower) * x using only the lower triangle of the row-major H
// (selfadjointView<Eigen::Lower>, solver.hpp:260, utils.hpp:466-467).
void
symv_lower(const double* H, isize n, const double* x, double* y)
{
  for (isize i = 0; i < n; ++i)
    y[i] = 0;
  for (isize i = 0; i < n; ++i) {
    const double* row = H + i * n;
    double acc = 0;
    double xi = x[i];
    for (isize j = 0; j < i; ++j) {
      acc += row[j] * x[j];
      y[j] += row[j] * xi;
    }
    y[i] += acc + row[i] * xi;
  }
}

// solver.hpp:243-318
void
iterative_residual(QP& qp, isize inner_pb_dim)
{
  Workspace& w = qp.work;
  Results& r = qp.results;
  const isize n = qp.model.dim, n_eq = qp.model.n_eq, n_in = qp.model.n_in;
  const isize nc = qp.n_constraints();
  ++qp.n_residuals;
  double* Hdx = w.Hdx.data();
  double* Adx = w.Adx.data();
  double* ATdy = w.CTz.data();
  double* err = w.err.data();
  const double* dw = w.dw_aug.data();
  for (isize i = 0; i < inner_pb_dim; ++i)
    err[i] = w.rhs[size_t(i)];
  switch (qp.hessian_type) {
    case PQP_HESSIAN_ZERO:
      break;
    case PQP_HESSIAN_DENSE:
      symv_lower(w.H_scaled.data(), n, dw, Hdx);
      for (isize i = 0; i < n; ++i)
        err[i] -= Hdx[i];
      break;
    case PQP_HESSIAN_DIAGONAL:
      for (isize i = 0; i < n; ++i) {
        Hdx[i] = w.H_scaled[size_t(i * n + i)] * dw[i];
        err[i] -= Hdx[i];
      }
      break;
  }
  for (isize i = 0; i < n; ++i)
    err[i] -= r.info.rho * dw[i];
  for (isize k = 0; k < n; ++k)
    ATdy[k] = 0;
  for (isize i = 0; i < n_eq; ++i) {
    double yi = dw[n + i];
    const double* row = w.A_scaled.data() + i * n;
    for (isize k = 0; k < n; ++k)
      ATdy[k] += row[k] * yi;
  }
  for (isize k = 0; k < n; ++k)
    err[k] -= ATdy[k];
  if (nc > n_in) {
    for (isize k = 0; k < n; ++k)
      w.active_part_z[size_t(n_in + k)] = dw[k] * w.i_scaled[size_t(k)];
  }
  for (isize i = 0; i < nc; i++) {
    isize j = w.current_bijection_map[size_t(i)];
    if (j < w.n_c) {
      if (i >= n_in) {
        err[i - n_in] -= dw[n + n_eq + j] * w.i_scaled[size_t(i - n_in)];
        err[n + n_eq + j] -= (w.active_part_z[size_t(i)] - dw[n + n_eq + j] * r.info.mu_in);
      } else {
        const double* row = w.C_scaled.data() + i * n;
        double dz = dw[n + n_eq + j];
        for (isize k = 0; k < n; ++k)
          err[k] -= dz * row[k];
        err[n + n_eq + j] -= (dot(row, dw, n) - dz * r.info.mu_in);
      }
    }
  }
  for (isize i = 0; i < n_eq; ++i) {
    Adx[i] = dot(w.A_scaled.data() + i * n, dw, n);
    err[n + i] -= Adx[i];
    err[n + i] += dw[n + i] * r.info.mu_eq;
  }
  qp.counters.level2_flops += 2.0 * (double(n) * double(n) + 2.0 * double(n_eq) * double(n) +
                                     2.0 * double(w.n_c) * double(n));
}

// solver.hpp:320-392 (PrimalDualLDLT)
void
solve_linear_system(QP& qp, double* dw, isize inner_pb_dim)
{
  ++qp.n_solves;
  qp.work.ldl.solve_in_place(dw, inner_pb_dim, qp.work.solve_work);
}

// solver.hpp:406-541
void
iterative_solve_with_permut_fact(QP& qp, double eps, isize inner_pb_dim)
{
  Workspace& w = qp.work;
  zero(w.err);
  isize it = 0;
  isize it_stability = 0;
  auto errn = [&]() { return infty_norm(w.err.data(), inner_pb_dim); };
  auto refine = [&]() {
    for (isize i = 0; i < inner_pb_dim; ++i)
      w.dw_aug[size_t(i)] = w.rhs[size_t(i)];
    solve_linear_system(qp, w.dw_aug.data(), inner_pb_dim);
    iterative_residual(qp, inner_pb_dim);
    ++it;
    double preverr = errn();
    while (errn() >= eps) {
      if (it >= qp.settings.nb_iterative_refinement)
        break;
      ++it;
      solve_linear_system(qp, w.err.data(), inner_pb_dim);
      for (isize i = 0; i < inner_pb_dim; ++i)
        w.dw_aug[size_t(i)] += w.err[size_t(i)];
      for (isize i = 0; i < inner_pb_dim; ++i)
        w.err[size_t(i)] = 0;
      iterative_residual(qp, inner_pb_dim);
      if (errn() > preverr)
        it_stability += 1;
      else
        it_stability = 0;
      if (it_stability == 2)
        break;
      preverr = errn();
    }
  };
  refine();
  if (errn() >= std::max(eps, qp.settings.eps_refact)) {
    refactorize(qp, qp.results.info.rho);
    it = 0;
    it_stability = 0;
    refine();
  }
  qp.results.info.iterative_residual = errn();
  for (isize i = 0; i < inner_pb_dim; ++i)
    w.rhs[size_t(i)] = 0;
}

// helpers.hpp:199-228
void
compute_equality_constrained_initial_guess(QP& qp)
{
  Workspace& w = qp.work;
  const isize n = qp.model.dim, n_eq = qp.model.n_eq;
  zero(w.rhs);
  for (isize i = 0; i < n; ++i)
    w.rhs[size_t(i)] = -w.g_scaled[size_t(i)];
  for (isize i = 0; i < n_eq; ++i)
    w.rhs[size_t(n + i)] = w.b_scaled[size_t(i)];
  iterative_solve_with_permut_fact(qp, 1.0, n + n_eq);
  for (isize i = 0; i < n; ++i)
    qp.results.x[size_t(i)] = w.dw_aug[size_t(i)];
  for (isize i = 0; i < n_eq; ++i)
    qp.results.y[size_t(i)] = w.dw_aug[size_t(n + i)];
  zero(w.dw_aug);
  zero(w.rhs);
}

// solver.hpp:564-614
void
bcl_update(QP& qp, double& primal_feasibility_lhs_new, double& bcl_eta_ext, double& bcl_eta_in,
           double bcl_eta_ext_init, double eps_in_min, double& new_bcl_mu_in,
           double& new_bcl_mu_eq, double& new_bcl_mu_in_inv, double& new_bcl_mu_eq_inv)
{
  const pqp_settings& s = qp.settings;
  Results& r = qp.results;
  if (primal_feasibility_lhs_new <= bcl_eta_ext || r.info.iter > s.safe_guard) {
    bcl_eta_ext *= std::pow(r.info.mu_in, s.beta_bcl);
    bcl_eta_in = std::max(bcl_eta_in * r.info.mu_in, eps_in_min);
  } else {
    r.y = qp.work.y_prev;
    r.z = qp.work.z_prev;
    new_bcl_mu_in = std::max(r.info.mu_in * s.mu_update_factor, s.mu_min_in);
    new_bcl_mu_eq = std::max(r.info.mu_eq * s.mu_update_factor, s.mu_min_eq);
    new_bcl_mu_in_inv = std::min(r.info.mu_in_inv * s.mu_update_inv_factor, s.mu_max_in_inv);
    new_bcl_mu_eq_inv = std::min(r.info.mu_eq_inv * s.mu_update_inv_factor, s.mu_max_eq_inv);
    bcl_eta_ext = bcl_eta_ext_init * std::pow(new_bcl_mu_in, s.alpha_bcl);
    bcl_eta_in = std::max(new_bcl_mu_in, eps_in_min);
  }
}
// solver.hpp:637-677
void
Martinez_update(QP& qp, double& primal_feasibility_lhs_new, double& primal_feasibility_lhs_old,
                double& bcl_eta_in, double eps_in_min, double& new_bcl_mu_in,
                double& new_bcl_mu_eq, double& new_bcl_mu_in_inv, double& new_bcl_mu_eq_inv)
{
  const pqp_settings& s = qp.settings;
  Results& r = qp.results;
  bcl_eta_in = std::max(bcl_eta_in * 0.1, eps_in_min);
  if (primal_feasibility_lhs_new <= 0.95 * primal_feasibility_lhs_old) {
  } else {
    new_bcl_mu_in = std::max(r.info.mu_in * s.mu_update_factor, s.mu_min_in);
    new_bcl_mu_eq = std::max(r.info.mu_eq * s.mu_update_factor, s.mu_min_eq);
    new_bcl_mu_in_inv = std::min(r.info.mu_in_inv * s.mu_update_inv_factor, s.mu_max_in_inv);
    new_bcl_mu_eq_inv = std::min(r.info.mu_eq_inv * s.mu_update_inv_factor, s.mu_max_eq_inv);
  }
}

// solver.hpp:687-743
double
compute_inner_loop_saddle_point(QP& qp)
{
  Workspace& w = qp.work;
  Results& r = qp.results;
  const isize nc = qp.n_constraints();
  const isize n = qp.model.dim, n_eq = qp.model.n_eq;
  double factor = (qp.settings.merit_function_type == PQP_MERIT_GPDAL) ? qp.settings.alpha_gpdal : 1.0;
  for (isize i = 0; i < nc; ++i) {
    double up = w.primal_residual_in_scaled_up[size_t(i)];
    double lo = r.si[size_t(i)];
    double v = (up > 0 ? up : 0.0) + (lo < 0 ? lo : 0.0);
    if (qp.settings.merit_function_type == PQP_MERIT_GPDAL)
      v -= factor * r.z[size_t(i)] * r.info.mu_in;
    else
      v -= r.z[size_t(i)] * r.info.mu_in;
    w.active_part_z[size_t(i)] = v;
  }
  double err = infty_norm(w.active_part_z.data(), nc);
  for (isize i = 0; i < n_eq; ++i)
    w.err[size_t(n + i)] = r.se[size_t(i)];
  double prim_eq_e = infty_norm(w.err.data() + n, n_eq);
  err = std::max(err, prim_eq_e);
  double dual_e = infty_norm(w.dual_residual_scaled.data(), n);
  err = std::max(err, dual_e);
  return err;
}

// ---------------------------------------------------------------- linesearch.hpp
// linesearch.hpp:549-786 (PrimalDualLDLT)
void
active_set_change(QP& qp)
{
  Workspace& w = qp.work;
  Results& r = qp.results;
  const isize n = qp.model.dim, n_eq = qp.model.n_eq, n_in = qp.model.n_in;
  const isize nc = qp.n_constraints();
  isize n_c_f = w.n_c;
  w.new_bijection_map = w.current_bijection_map;
  {
    std::vector<isize> planned_to_delete(size_t(std::max<isize>(nc, 1)));
    isize planned_to_delete_count = 0;
    for (isize i = 0; i < nc; i++) {
      if (w.current_bijection_map[size_t(i)] < w.n_c) {
        if (!w.active_inequalities[size_t(i)]) {
          planned_to_delete[size_t(planned_to_delete_count)] =
            w.current_bijection_map[size_t(i)] + n + n_eq;
          ++planned_to_delete_count;
          for (isize j = 0; j < nc; j++) {
            if (w.new_bijection_map[size_t(j)] > w.new_bijection_map[size_t(i)])
              w.new_bijection_map[size_t(j)] -= 1;
          }
          n_c_f -= 1;
          w.new_bijection_map[size_t(i)] = nc - 1;
        }
      }
    }
    std::sort(planned_to_delete.begin(), planned_to_delete.begin() + planned_to_delete_count);
    w.ldl.delete_at(planned_to_delete.data(), planned_to_delete_count);
    qp.n_deleted += planned_to_delete_count;
    if (planned_to_delete_count > 0)
      w.constraints_changed = true;
  }
  {
    std::vector<isize> planned_to_add(size_t(std::max<isize>(nc, 1)));
    isize planned_to_add_count = 0;
    double mu_in_neg = -r.info.mu_in;
    isize n_c = n_c_f;
    for (isize i = 0; i < nc; i++) {
      if (w.active_inequalities[size_t(i)]) {
        if (w.new_bijection_map[size_t(i)] >= n_c_f) {
          planned_to_add[size_t(planned_to_add_count)] = i;
          ++planned_to_add_count;
          for (isize j = 0; j < nc; j++) {
            if (w.new_bijection_map[size_t(j)] < w.new_bijection_map[size_t(i)] &&
                w.new_bijection_map[size_t(j)] >= n_c_f)
              w.new_bijection_map[size_t(j)] += 1;
          }
          w.new_bijection_map[size_t(i)] = n_c_f;
          n_c_f += 1;
        }
      }
    }
    {
      isize rows = n + n_eq + n_c_f;
      Vec new_cols(size_t(rows * std::max<isize>(planned_to_add_count, 1)), 0.0);
      for (isize k = 0; k < planned_to_add_count; ++k) {
        isize index = planned_to_add[size_t(k)];
        double* col = new_cols.data() + k * rows;
        if (index >= n_in) {
          for (isize t = 0; t < n; ++t)
            col[t] = 0;
          col[index - n_in] = w.i_scaled[size_t(index - n_in)];
        } else {
          for (isize t = 0; t < n; ++t)
            col[t] = w.C_scaled[size_t(index * n + t)];
        }
        for (isize t = n; t < rows; ++t)
          col[t] = 0;
        col[n + n_eq + n_c + k] = mu_in_neg;
      }
      w.ldl.insert_block_at(n + n_eq + n_c, new_cols.data(), rows, planned_to_add_count);
      qp.n_inserted += planned_to_add_count;
    }
    if (planned_to_add_count > 0)
      w.constraints_changed = true;
  }
  w.n_c = n_c_f;
  w.current_bijection_map = w.new_bijection_map;
}

struct DerivRes
{
  double a, b, grad;
};

// linesearch.hpp:49-167 (GPDAL) and :178-311 (PDAL)
DerivRes
derivative_results(QP& qp, double alpha)
{
  Workspace& w = qp.work;
  Results& r = qp.results;
  const pqp_settings& s = qp.settings;
  const isize n = qp.model.dim, n_eq = qp.model.n_eq;
  const isize nc = qp.n_constraints();
  const bool gpdal = s.merit_function_type == PQP_MERIT_GPDAL;
  ++qp.n_ls_evals;
  const double* dx = w.dw_aug.data();
  const double* dy = w.dw_aug.data() + n;
  const double* dz = w.dw_aug.data() + n + n_eq;
  for (isize i = 0; i < nc; ++i) {
    w.primal_residual_in_scaled_up_plus_alphaCdx[size_t(i)] =
      w.primal_residual_in_scaled_up[size_t(i)] + w.Cdx[size_t(i)] * alpha;
    w.primal_residual_in_scaled_low_plus_alphaCdx[size_t(i)] =
      r.si[size_t(i)] + w.Cdx[size_t(i)] * alpha;
  }
  double a = dot(dx, w.Hdx.data(), n) + r.info.mu_eq_inv * dot(w.Adx.data(), w.Adx.data(), n_eq) +
             r.info.rho * dot(dx, dx, n);
  double* err_eq = w.err.data() + n;
  for (isize i = 0; i < n_eq; ++i)
    err_eq[i] = w.Adx[size_t(i)] - dy[i] * r.info.mu_eq;
  if (gpdal)
    a += dot(err_eq, err_eq, n_eq) * r.info.mu_eq_inv;
  else
    a += dot(err_eq, err_eq, n_eq) * r.info.mu_eq_inv * r.info.nu;
  for (isize i = 0; i < n; ++i)
    w.err[size_t(i)] = r.info.rho * (r.x[size_t(i)] - w.x_prev[size_t(i)]) + w.g_scaled[size_t(i)];
  double tmp = 0;
  for (isize i = 0; i < n_eq; ++i)
    tmp += w.Adx[size_t(i)] * (r.se[size_t(i)] + r.y[size_t(i)] * r.info.mu_eq);
  double b = dot(r.x.data(), w.Hdx.data(), n) + dot(w.err.data(), dx, n) + r.info.mu_eq_inv * tmp;
  for (isize i = 0; i < n_eq; ++i)
    w.rhs[size_t(n + i)] = r.se[size_t(i)];
  if (gpdal)
    b += r.info.mu_eq_inv * dot(err_eq, w.rhs.data() + n, n_eq);
  else
    b += r.info.nu * r.info.mu_eq_inv * dot(err_eq, w.rhs.data() + n, n_eq);
  double* err_in = w.err.data() + n + n_eq;
  for (isize i = 0; i < nc; ++i) {
    bool up = w.primal_residual_in_scaled_up_plus_alphaCdx[size_t(i)] > 0.;
    bool lo = w.primal_residual_in_scaled_low_plus_alphaCdx[size_t(i)] < 0.;
    err_in[i] = (up || lo) ? w.Cdx[size_t(i)] : 0.0;
    w.active_part_z[size_t(i)] =
      (up ? w.primal_residual_in_scaled_up[size_t(i)] : 0.0) + (lo ? r.si[size_t(i)] : 0.0);
  }
  if (gpdal) {
    a += r.info.mu_in_inv * dot(err_in, err_in, nc) / s.alpha_gpdal;
    a += r.info.mu_in * (1. - s.alpha_gpdal) * dot(dz, dz, nc);
    b += r.info.mu_in_inv * dot(w.active_part_z.data(), err_in, nc) / s.alpha_gpdal;
    b += r.info.mu_in * (1. - s.alpha_gpdal) * dot(dz, r.z.data(), nc);
  } else {
    a += r.info.mu_in_inv * dot(err_in, err_in, nc);
    b += r.info.mu_in_inv * dot(w.active_part_z.data(), err_in, nc);
    for (isize i = 0; i < nc; ++i) {
      err_in[i] -= dz[i] * r.info.mu_in;
      w.active_part_z[size_t(i)] -= r.z[size_t(i)] * r.info.mu_in;
    }
    a += r.info.nu * r.info.mu_in_inv * dot(err_in, err_in, nc);
    b += r.info.nu * r.info.mu_in_inv * dot(err_in, w.active_part_z.data(), nc);
  }
  qp.counters.level2_flops += 10.0 * double(n + n_eq + nc);
  return { a, b, a * alpha + b };
}

// linesearch.hpp:320-538
void
primal_dual_ls(QP& qp)
{
  Workspace& w = qp.work;
  Results& r = qp.results;
  const isize nc = qp.n_constraints();
  w.alpha = 1;
  double alpha_ = 1.;
  w.alphas.clear();
  for (isize i = 0; i < nc; i++) {
    if (w.Cdx[size_t(i)] != 0.) {
      alpha_ = -w.primal_residual_in_scaled_up[size_t(i)] / (w.Cdx[size_t(i)] + MACHINE_EPS);
      if (alpha_ > MACHINE_EPS)
        w.alphas.push_back(alpha_);
      alpha_ = -r.si[size_t(i)] / (w.Cdx[size_t(i)] + MACHINE_EPS);
      if (alpha_ > MACHINE_EPS)
        w.alphas.push_back(alpha_);
    }
  }
  std::sort(w.alphas.begin(), w.alphas.end());
  w.alphas.erase(std::unique(w.alphas.begin(), w.alphas.end()), w.alphas.end());
  isize n_alpha = isize(w.alphas.size());
  if (n_alpha == 0) {
    DerivRes res = derivative_results(qp, 0.0);
    w.alpha = -res.b / res.a;
    return;
  }
  const double infty = std::numeric_limits<double>::infinity();
  double last_neg_grad = 0;
  double alpha_last_neg = 0;
  double first_pos_grad = 0;
  double alpha_first_pos = infty;
  for (isize i = 0; i < n_alpha; ++i) {
    alpha_ = w.alphas[size_t(i)];
    double gr = derivative_results(qp, alpha_).grad;
    if (gr < 0) {
      alpha_last_neg = alpha_;
      last_neg_grad = gr;
    } else {
      first_pos_grad = gr;
      alpha_first_pos = alpha_;
      break;
    }
  }
  if (alpha_last_neg == 0.0)
    last_neg_grad = derivative_results(qp, alpha_last_neg).grad;
  if (alpha_first_pos == infty) {
    DerivRes res = derivative_results(qp, 2 * alpha_last_neg + 1);
    w.alpha = -res.b / res.a;
  } else {
    w.alpha = std::fabs(alpha_last_neg - last_neg_grad * (alpha_first_pos - alpha_last_neg) /
                                           (first_pos_grad - last_neg_grad));
  }
}

// ------------------------------------------------------------------- utils.hpp
// utils.hpp:164-252
void
global_primal_residual(QP& qp, double& primal_feasibility_lhs, double& primal_feasibility_eq_rhs_0,
                       double& primal_feasibility_in_rhs_0, double& primal_feasibility_eq_lhs,
                       double& primal_feasibility_in_lhs)
{
  Workspace& w = qp.work;
  Results& r = qp.results;
  Model& m = qp.model;
  Scaler sc(qp);
  const isize n = m.dim, n_eq = m.n_eq, n_in = m.n_in;
  const isize nc = qp.n_constraints();
  for (isize i = 0; i < n_eq; ++i)
    r.se[size_t(i)] = dot(w.A_scaled.data() + i * n, r.x.data(), n);
  for (isize i = 0; i < n_in; ++i)
    w.primal_residual_in_scaled_up[size_t(i)] = dot(w.C_scaled.data() + i * n, r.x.data(), n);
  if (qp.box_constraints) {
    for (isize i = 0; i < n; ++i)
      w.primal_residual_in_scaled_up[size_t(n_in + i)] = r.x[size_t(i)];
    sc.unscale_primal(w.primal_residual_in_scaled_up.data() + n_in);
  }
  sc.unscale_primal_residual_eq(r.se.data());
  primal_feasibility_eq_rhs_0 = infty_norm(r.se.data(), n_eq);
  sc.unscale_primal_residual_in(w.primal_residual_in_scaled_up.data());
  primal_feasibility_in_rhs_0 = infty_norm(w.primal_residual_in_scaled_up.data(), n_in);
  for (isize i = 0; i < n_in; ++i) {
    double v = w.primal_residual_in_scaled_up[size_t(i)];
    double pu = v - m.u[size_t(i)];
    double pl = v - m.l[size_t(i)];
    r.si[size_t(i)] = (pu > 0 ? pu : 0.0) + (pl < 0 ? pl : 0.0);
  }
  if (qp.box_constraints) {
    for (isize i = 0; i < n; ++i) {
      double v = w.primal_residual_in_scaled_up[size_t(n_in + i)];
      double pu = v - m.u_box[size_t(i)];
      double pl = v - m.l_box[size_t(i)];
      r.si[size_t(n_in + i)] = (pu > 0 ? pu : 0.0) + (pl < 0 ? pl : 0.0);
      w.active_part_z[size_t(n_in + i)] = r.x[size_t(i)] - r.si[size_t(n_in + i)];
    }
    primal_feasibility_in_rhs_0 =
      std::max(primal_feasibility_in_rhs_0, infty_norm(w.active_part_z.data() + n_in, n));
    primal_feasibility_in_rhs_0 = std::max(primal_feasibility_in_rhs_0, infty_norm(r.x.data(), n));
  }
  for (isize i = 0; i < n_eq; ++i)
    r.se[size_t(i)] -= m.b[size_t(i)];
  primal_feasibility_in_lhs = infty_norm(r.si.data(), nc);
  primal_feasibility_eq_lhs = infty_norm(r.se.data(), n_eq);
  primal_feasibility_lhs = std::max(primal_feasibility_eq_lhs, primal_feasibility_in_lhs);
  if (qp.settings.primal_infeasibility_solving && r.info.status == PQP_PRIMAL_INFEASIBLE) {
    for (isize k = 0; k < n; ++k)
      w.rhs[size_t(k)] = 0;
    for (isize i = 0; i < n_eq; ++i)
      for (isize k = 0; k < n; ++k)
        w.rhs[size_t(k)] += m.A[size_t(i * n + k)] * r.se[size_t(i)];
    for (isize i = 0; i < n_in; ++i)
      for (isize k = 0; k < n; ++k)
        w.rhs[size_t(k)] += m.C[size_t(i * n + k)] * r.si[size_t(i)];
    primal_feasibility_lhs = infty_norm(w.rhs.data(), n);
  }
  sc.scale_primal_residual_eq(r.se.data());
  qp.counters.level2_flops += 2.0 * double(n_eq + n_in) * double(n);
}

// utils.hpp:269-324.  Mutates ATdy, CTdz, dy, dz in place.
bool
global_primal_residual_infeasibility(QP& qp, double* ATdy, double* CTdz, double* dy, double* dz)
{
  Workspace& w = qp.work;
  Scaler sc(qp);
  const isize n = qp.model.dim, n_eq = qp.model.n_eq, n_in = qp.model.n_in;
  const isize nc = qp.n_constraints();
  bool res = infty_norm(dy, n_eq) != 0 || infty_norm(dz, nc) != 0;
  if (!res)
    return res;
  sc.unscale_dual_residual(ATdy);
  sc.unscale_dual_residual(CTdz);
  double lower_bound_1 = dot(dy, w.b_scaled.data(), n_eq);
  for (isize i = 0; i < n_in; ++i) {
    double pz = dz[i] > 0 ? dz[i] : 0.0;
    double nz = dz[i] < 0 ? dz[i] : 0.0;
    lower_bound_1 += pz * w.u_scaled[size_t(i)];
    lower_bound_1 -= nz * w.l_scaled[size_t(i)];
  }
  sc.unscale_dual_eq(dy);
  sc.unscale_dual_in(dz);
  if (qp.box_constraints) {
    for (isize i = 0; i < n; ++i) {
      double v = dz[n_in + i];
      double pz = v > 0 ? v : 0.0;
      double nz = v < 0 ? v : 0.0;
      lower_bound_1 += pz * w.u_box_scaled[size_t(i)];
      lower_bound_1 -= nz * w.l_box_scaled[size_t(i)];
    }
    sc.unscale_box_dual_in(dz + n_in);
  }
  double upper_bound =
    qp.settings.eps_primal_inf * std::max(infty_norm(dy, n_eq), infty_norm(dz, nc));
  double lower_bound_2 = 0;
  for (isize k = 0; k < n; ++k)
    lower_bound_2 = std::max(lower_bound_2, std::fabs(ATdy[k] + CTdz[k]));
  res = lower_bound_2 <= upper_bound && lower_bound_1 <= -upper_bound;
  if (std::getenv("PQO_TRACE_CERT"))
    std::fprintf(stderr, "[oracle]   primal-inf cert: lb2 %.4e <= ub %.4e ? lb1 %.4e <= %.4e ? -> %d\n", lower_bound_2,
                 upper_bound, lower_bound_1, -upper_bound, int(res));
  return res;
}

// utils.hpp:343-419.  Mutates Adx, Cdx, Hdx, dx in place.
bool
global_dual_residual_infeasibility(QP& qp, double* Adx, double* Cdx, double* Hdx, double* dx)
{
  Workspace& w = qp.work;
  Scaler sc(qp);
  const isize n = qp.model.dim, n_eq = qp.model.n_eq, n_in = qp.model.n_in;
  sc.unscale_dual_residual(Hdx);
  sc.unscale_primal_residual_eq(Adx);
  sc.unscale_primal_residual_in(Cdx);
  if (qp.box_constraints)
    sc.unscale_box_primal_residual_in(Cdx + n_in);
  double gdx = dot(dx, w.g_scaled.data(), n);
  sc.unscale_primal(dx);
  double bound = infty_norm(dx, n) * qp.settings.eps_dual_inf;
  double bound_neg = -bound;
  bool first_cond = infty_norm(Adx, n_eq) <= bound;
  for (isize iter = 0; iter < n_in; ++iter) {
    double Cdx_i = Cdx[iter];
    if (w.u_scaled[size_t(iter)] <= 1.E20 && w.l_scaled[size_t(iter)] >= -1.E20) {
      first_cond = first_cond && Cdx_i <= bound && Cdx_i >= bound_neg;
    } else if (w.u_scaled[size_t(iter)] > 1.E20) {
      first_cond = first_cond && Cdx_i >= bound_neg;
    } else if (w.l_scaled[size_t(iter)] < -1.E20) {
      first_cond = first_cond && Cdx_i <= bound;
    }
  }
  if (qp.box_constraints) {
    for (isize iter = 0; iter < n; ++iter) {
      double dx_i = dx[iter];
      if (w.u_box_scaled[size_t(iter)] <= 1.E20 && w.l_box_scaled[size_t(iter)] >= -1.E20) {
        first_cond = first_cond && dx_i <= bound && dx_i >= bound_neg;
      } else if (w.u_box_scaled[size_t(iter)] > 1.E20) {
        first_cond = first_cond && dx_i >= bound_neg;
      } else if (w.l_box_scaled[size_t(iter)] < -1.E20) {
        first_cond = first_cond && dx_i <= bound;
      }
    }
  }
  bound *= qp.ruiz.c;
  bound_neg *= qp.ruiz.c;
  bool second_cond_alt1 = infty_norm(Hdx, n) <= bound && gdx <= bound_neg;
  bool res = first_cond && second_cond_alt1 && infty_norm(dx, n) != 0;
  return res;
}

// utils.hpp:437-587
void
global_dual_residual(QP& qp, double& dual_feasibility_lhs, double& dual_feasibility_rhs_0,
                     double& dual_feasibility_rhs_1, double& dual_feasibility_rhs_3,
                     double& rhs_duality_gap, double& duality_gap)
{
  Workspace& w = qp.work;
  Results& r = qp.results;
  Model& m = qp.model;
  Scaler sc(qp);
  const isize n = m.dim, n_eq = m.n_eq, n_in = m.n_in;
  double* CTz = w.CTz.data();
  w.dual_residual_scaled = w.g_scaled;
  switch (qp.hessian_type) {
    case PQP_HESSIAN_ZERO:
      dual_feasibility_rhs_0 = 0;
      break;
    case PQP_HESSIAN_DENSE:
      symv_lower(w.H_scaled.data(), n, r.x.data(), CTz);
      for (isize k = 0; k < n; ++k)
        w.dual_residual_scaled[size_t(k)] += CTz[k];
      sc.unscale_dual_residual(CTz);
      dual_feasibility_rhs_0 = infty_norm(CTz, n);
      break;
    case PQP_HESSIAN_DIAGONAL:
      for (isize k = 0; k < n; ++k) {
        CTz[k] = w.H_scaled[size_t(k * n + k)] * r.x[size_t(k)];
        w.dual_residual_scaled[size_t(k)] += CTz[k];
      }
      sc.unscale_dual_residual(CTz);
      dual_feasibility_rhs_0 = infty_norm(CTz, n);
      break;
  }
  sc.unscale_primal(r.x.data());
  duality_gap = dot(m.g.data(), r.x.data(), n);
  rhs_duality_gap = std::fabs(duality_gap);
  if (qp.hessian_type != PQP_HESSIAN_ZERO) {
    double xHx = dot(CTz, r.x.data(), n);
    duality_gap += xHx;
    rhs_duality_gap = std::max(rhs_duality_gap, std::fabs(xHx));
  }
  sc.scale_primal(r.x.data());

  for (isize k = 0; k < n; ++k)
    CTz[k] = 0;
  for (isize i = 0; i < n_eq; ++i) {
    double yi = r.y[size_t(i)];
    const double* row = w.A_scaled.data() + i * n;
    for (isize k = 0; k < n; ++k)
      CTz[k] += row[k] * yi;
  }
  for (isize k = 0; k < n; ++k)
    w.dual_residual_scaled[size_t(k)] += CTz[k];
  sc.unscale_dual_residual(CTz);
  dual_feasibility_rhs_1 = infty_norm(CTz, n);

  for (isize k = 0; k < n; ++k)
    CTz[k] = 0;
  for (isize i = 0; i < n_in; ++i) {
    double zi = r.z[size_t(i)];
    const double* row = w.C_scaled.data() + i * n;
    for (isize k = 0; k < n; ++k)
      CTz[k] += row[k] * zi;
  }
  for (isize k = 0; k < n; ++k)
    w.dual_residual_scaled[size_t(k)] += CTz[k];
  sc.unscale_dual_residual(CTz);
  dual_feasibility_rhs_3 = infty_norm(CTz, n);
  if (qp.box_constraints) {
    for (isize k = 0; k < n; ++k) {
      CTz[k] = r.z[size_t(n_in + k)] * w.i_scaled[size_t(k)];
      w.dual_residual_scaled[size_t(k)] += CTz[k];
    }
    sc.unscale_dual_residual(CTz);
    dual_feasibility_rhs_3 = std::max(infty_norm(CTz, n), dual_feasibility_rhs_3);
  }
  sc.unscale_dual_residual(w.dual_residual_scaled.data());
  dual_feasibility_lhs = infty_norm(w.dual_residual_scaled.data(), n);
  sc.scale_dual_residual(w.dual_residual_scaled.data());

  sc.unscale_dual_eq(r.y.data());
  const double by = dot(m.b.data(), r.y.data(), n_eq);
  rhs_duality_gap = std::max(rhs_duality_gap, std::fabs(by));
  duality_gap += by;
  sc.scale_dual_eq(r.y.data());

  sc.unscale_dual_in(r.z.data());
  const double ib = infinite_bound();
  double zu = 0, zl = 0;
  for (isize i = 0; i < n_in; ++i) {
    double zi = r.z[size_t(i)];
    zu += (w.active_set_up[size_t(i)] ? zi : 0.0) * (m.u[size_t(i)] < ib ? m.u[size_t(i)] : ib);
    zl += (w.active_set_low[size_t(i)] ? zi : 0.0) * (m.l[size_t(i)] > -ib ? m.l[size_t(i)] : -ib);
  }
  rhs_duality_gap = std::max(rhs_duality_gap, std::fabs(zu));
  duality_gap += zu;
  rhs_duality_gap = std::max(rhs_duality_gap, std::fabs(zl));
  duality_gap += zl;
  sc.scale_dual_in(r.z.data());
  if (qp.box_constraints) {
    sc.unscale_box_dual_in(r.z.data() + n_in);
    zu = 0;
    zl = 0;
    for (isize i = 0; i < n; ++i) {
      double zi = r.z[size_t(n_in + i)];
      zu += (w.active_set_up[size_t(n_in + i)] ? zi : 0.0) *
            (m.u_box[size_t(i)] < ib ? m.u_box[size_t(i)] : ib);
      zl += (w.active_set_low[size_t(n_in + i)] ? zi : 0.0) *
            (m.l_box[size_t(i)] > -ib ? m.l_box[size_t(i)] : -ib);
    }
    rhs_duality_gap = std::max(rhs_duality_gap, std::fabs(zu));
    duality_gap += zu;
    rhs_duality_gap = std::max(rhs_duality_gap, std::fabs(zl));
    duality_gap += zl;
    sc.scale_box_dual_in(r.z.data() + n_in);
  }
  qp.counters.level2_flops += 2.0 * (double(n) * double(n) + double(n_eq + n_in) * double(n));
}

// solver.hpp:754-869
void
primal_dual_semi_smooth_newton_step(QP& qp, double eps)
{
  Workspace& w = qp.work;
  Results& r = qp.results;
  const pqp_settings& s = qp.settings;
  const isize n = qp.model.dim, n_eq = qp.model.n_eq, n_in = qp.model.n_in;
  const isize nc = qp.n_constraints();
  isize numactive_inequalities = 0;
  for (isize i = 0; i < nc; ++i) {
    w.active_set_up[size_t(i)] = w.primal_residual_in_scaled_up[size_t(i)] >= 0;
    w.active_set_low[size_t(i)] = r.si[size_t(i)] <= 0;
    w.active_inequalities[size_t(i)] = w.active_set_up[size_t(i)] || w.active_set_low[size_t(i)];
    numactive_inequalities += w.active_inequalities[size_t(i)] ? 1 : 0;
  }
  isize inner_pb_dim = n + n_eq + numactive_inequalities;
  zero(w.rhs);
  zero(w.dw_aug);
  active_set_change(qp);
  for (isize k = 0; k < n; ++k)
    w.rhs[size_t(k)] = -w.dual_residual_scaled[size_t(k)];
  if (qp.box_constraints)
    for (isize k = 0; k < n; ++k)
      w.active_part_z[size_t(n_in + k)] = r.z[size_t(n_in + k)] * w.i_scaled[size_t(k)];
  for (isize i = 0; i < n_eq; ++i)
    w.rhs[size_t(n + i)] = -r.se[size_t(i)];
  const double zfac = (s.merit_function_type == PQP_MERIT_GPDAL) ? s.alpha_gpdal : 1.0;
  for (isize i = 0; i < nc; i++) {
    isize j = w.current_bijection_map[size_t(i)];
    if (j < w.n_c) {
      if (w.active_set_up[size_t(i)]) {
        if (s.merit_function_type == PQP_MERIT_GPDAL)
          w.rhs[size_t(j + n + n_eq)] =
            -w.primal_residual_in_scaled_up[size_t(i)] + r.z[size_t(i)] * r.info.mu_in * zfac;
        else
          w.rhs[size_t(j + n + n_eq)] =
            -w.primal_residual_in_scaled_up[size_t(i)] + r.z[size_t(i)] * r.info.mu_in;
      } else if (w.active_set_low[size_t(i)]) {
        if (s.merit_function_type == PQP_MERIT_GPDAL)
          w.rhs[size_t(j + n + n_eq)] = -r.si[size_t(i)] + r.z[size_t(i)] * r.info.mu_in * zfac;
        else
          w.rhs[size_t(j + n + n_eq)] = -r.si[size_t(i)] + r.z[size_t(i)] * r.info.mu_in;
      }
    } else {
      if (i >= n_in) {
        w.rhs[size_t(i - n_in)] += w.active_part_z[size_t(i)];
      } else {
        double zi = r.z[size_t(i)];
        const double* row = w.C_scaled.data() + i * n;
        for (isize k = 0; k < n; ++k)
          w.rhs[size_t(k)] += zi * row[k];
      }
    }
  }
  qp.counters.level2_flops += 2.0 * double(n_in) * double(n);
  iterative_solve_with_permut_fact(qp, eps, inner_pb_dim);
  for (isize j = 0; j < nc; ++j) {
    isize i = w.current_bijection_map[size_t(j)];
    if (i < w.n_c)
      w.active_part_z[size_t(j)] = w.dw_aug[size_t(n + n_eq + i)];
    else
      w.active_part_z[size_t(j)] = -r.z[size_t(j)];
  }
  for (isize j = 0; j < nc; ++j)
    w.dw_aug[size_t(n + n_eq + j)] = w.active_part_z[size_t(j)];
}

// solver.hpp:882-1077
void
primal_dual_newton_semi_smooth(QP& qp, double eps_int)
{
  Workspace& w = qp.work;
  Results& r = qp.results;
  const pqp_settings& s = qp.settings;
  const isize n = qp.model.dim, n_eq = qp.model.n_eq, n_in = qp.model.n_in;
  const isize nc = qp.n_constraints();
  const isize total = n + n_eq + nc;
  double err_in = 1.e6;
  Vec CTdz(static_cast<size_t>(n));
  for (std::int64_t iter = 0; iter <= s.max_iter_in; ++iter) {
    if (iter == s.max_iter_in) {
      r.info.iter += s.max_iter_in + 1;
      break;
    }
    primal_dual_semi_smooth_newton_step(qp, eps_int);
    double* Hdx = w.Hdx.data();
    double* Adx = w.Adx.data();
    double* Cdx = w.Cdx.data();
    double* ATdy = w.CTz.data();
    double* dx = w.dw_aug.data();
    double* dy = w.dw_aug.data() + n;
    double* dz = w.dw_aug.data() + n + n_eq;
    std::fill(CTdz.begin(), CTdz.end(), 0.0);
    if (n_in > 0) {
      for (isize i = 0; i < n_in; ++i) {
        const double* row = w.C_scaled.data() + i * n;
        Cdx[i] = dot(row, dx, n);
        double dzi = dz[i];
        for (isize k = 0; k < n; ++k)
          CTdz[size_t(k)] += row[k] * dzi;
      }
      qp.counters.level2_flops += 4.0 * double(n_in) * double(n);
    }
    if (qp.box_constraints) {
      for (isize k = 0; k < n; ++k) {
        w.active_part_z[size_t(n_in + k)] = dz[n_in + k] * w.i_scaled[size_t(k)];
        CTdz[size_t(k)] += w.active_part_z[size_t(n_in + k)];
        Cdx[n_in + k] = dx[k] * w.i_scaled[size_t(k)];
      }
    }
    if (s.merit_function_type == PQP_MERIT_GPDAL)
      for (isize i = 0; i < nc; ++i)
        Cdx[i] += (s.alpha_gpdal - 1.) * r.info.mu_in * dz[i];
    if (n_in > 0 || qp.box_constraints)
      primal_dual_ls(qp);
    double alpha = w.alpha;
    {
      double nrm = 0;
      for (isize i = 0; i < total; ++i)
        nrm = std::max(nrm, std::fabs(alpha * w.dw_aug[size_t(i)]));
      if (nrm < 1.E-11 && iter > 0) {
        r.info.iter += iter + 1;
        break;
      }
    }
    for (isize k = 0; k < n; ++k)
      r.x[size_t(k)] += alpha * dx[k];
    for (isize i = 0; i < nc; ++i) {
      w.primal_residual_in_scaled_up[size_t(i)] += alpha * Cdx[i];
      r.si[size_t(i)] += alpha * Cdx[i];
    }
    for (isize i = 0; i < n_eq; ++i) {
      r.se[size_t(i)] += alpha * (Adx[i] - r.info.mu_eq * dy[i]);
      r.y[size_t(i)] += alpha * dy[i];
    }
    for (isize i = 0; i < nc; ++i)
      r.z[size_t(i)] += alpha * dz[i];
    if (qp.hessian_type == PQP_HESSIAN_ZERO) {
      for (isize k = 0; k < n; ++k)
        w.dual_residual_scaled[size_t(k)] += alpha * (r.info.rho * dx[k] + ATdy[k] + CTdz[size_t(k)]);
    } else {
      for (isize k = 0; k < n; ++k)
        w.dual_residual_scaled[size_t(k)] +=
          alpha * (r.info.rho * dx[k] + Hdx[k] + ATdy[k] + CTdz[size_t(k)]);
    }
    err_in = compute_inner_loop_saddle_point(qp);
    if (s.verbose) { // solver.hpp:1021-1027
      const double rec[8] = { 2.0, double(iter + 1), err_in, alpha, 0.0, 0.0, 0.0, 0.0 };
      qp.trace.insert(qp.trace.end(), rec, rec + 8);
    }
    if (iter % s.frequence_infeasibility_check == 0 || s.primal_infeasibility_solving) {
      bool is_primal_infeasible =
        global_primal_residual_infeasibility(qp, ATdy, CTdz.data(), dy, dz);
      bool is_dual_infeasible = global_dual_residual_infeasibility(qp, Adx, Cdx, Hdx, dx);
      if (is_primal_infeasible) {
        r.info.status = PQP_PRIMAL_INFEASIBLE;
        if (!s.primal_infeasibility_solving) {
          r.info.iter += iter + 1;
          break;
        }
      } else if (is_dual_infeasible) {
        r.info.status = PQP_DUAL_INFEASIBLE;
        r.info.iter += iter + 1;
        break;
      }
    }
    if (err_in <= eps_int) {
      r.info.iter += iter + 1;
      break;
    }
  }
}

// reused in 5 places of qp_solve (solver.hpp:1231-1240 etc.)
void
active_set_from_z(QP& qp)
{
  Workspace& w = qp.work;
  const isize nc = qp.n_constraints();
  w.n_c = 0;
  for (isize i = 0; i < nc; i++)
    w.active_inequalities[size_t(i)] = qp.results.z[size_t(i)] != 0;
  active_set_change(qp);
}

void
scale_warm_start(QP& qp)
{
  Scaler sc(qp);
  sc.scale_primal(qp.results.x.data());
  sc.scale_dual_eq(qp.results.y.data());
  sc.scale_dual_in(qp.results.z.data());
  if (qp.box_constraints)
    sc.scale_box_dual_in(qp.results.z.data() + qp.model.n_in);
}

double
objective_value(const QP& qp)
{
  // solver.hpp:1771-1780: diagonal + strict lower (column tail) of model.H
  const Model& m = qp.model;
  const Vec& x = qp.results.x;
  const isize n = m.dim;
  double obj = 0;
  for (isize j = 0; j < n; ++j) {
    obj += 0.5 * (x[size_t(j)] * x[size_t(j)]) * m.H[size_t(j * n + j)];
    double acc = 0;
    for (isize i = j + 1; i < n; ++i)
      acc += m.H[size_t(i * n + j)] * x[size_t(i)];
    obj += x[size_t(j)] * acc;
  }
  obj += dot(m.g.data(), x.data(), n);
  return obj;
}

// solver.hpp:1088-1843
void
qp_solve(QP& qp)
{
  Workspace& w = qp.work;
  Results& r = qp.results;
  const pqp_settings& s = qp.settings;
  Model& m = qp.model;
  Scaler sc(qp);
  const isize n = m.dim, n_eq = m.n_eq, n_in = m.n_in;
  const isize nc = qp.n_constraints();
  auto t0 = std::chrono::steady_clock::now();
  qp.counters.reset();
  qp.trace.clear();
  w.ldl.ctr = &qp.counters;

  if (w.dirty) {
    switch (s.initial_guess) {
      case PQP_EQUALITY_CONSTRAINED_INITIAL_GUESS:
      case PQP_NO_INITIAL_GUESS:
        work_cleanup(w, nc);
        results_cleanup(r, s);
        break;
      case PQP_COLD_START_WITH_PREVIOUS_RESULT:
      case PQP_WARM_START:
        work_cleanup(w, nc);
        cold_start(r.info, s);
        scale_warm_start(qp);
        break;
      case PQP_WARM_START_WITH_PREVIOUS_RESULT:
        cleanup_statistics(r.info);
        scale_warm_start(qp);
        break;
    }
    if (s.initial_guess != PQP_WARM_START_WITH_PREVIOUS_RESULT) {
      copy_model_to_scaled(qp, false);
      setup_equilibration(qp, false);
      setup_factorization(qp);
    }
    switch (s.initial_guess) {
      case PQP_EQUALITY_CONSTRAINED_INITIAL_GUESS:
        compute_equality_constrained_initial_guess(qp);
        break;
      case PQP_COLD_START_WITH_PREVIOUS_RESULT:
      case PQP_WARM_START:
        active_set_from_z(qp);
        break;
      default:
        break;
    }
  } else {
    switch (s.initial_guess) {
      case PQP_EQUALITY_CONSTRAINED_INITIAL_GUESS:
        setup_factorization(qp);
        compute_equality_constrained_initial_guess(qp);
        break;
      case PQP_COLD_START_WITH_PREVIOUS_RESULT:
      case PQP_WARM_START:
        scale_warm_start(qp);
        setup_factorization(qp);
        active_set_from_z(qp);
        break;
      case PQP_NO_INITIAL_GUESS:
        setup_factorization(qp);
        break;
      case PQP_WARM_START_WITH_PREVIOUS_RESULT:
        scale_warm_start(qp);
        if (w.refactorize) {
          setup_factorization(qp);
          active_set_from_z(qp);
        }
        break;
    }
  }
  double bcl_eta_ext_init = std::pow(0.1, s.alpha_bcl);
  double bcl_eta_ext = bcl_eta_ext_init;
  double bcl_eta_in = 1;
  double eps_in_min = std::min(s.eps_abs, 1.E-9);

  double primal_feasibility_eq_rhs_0 = 0, primal_feasibility_in_rhs_0 = 0;
  double dual_feasibility_rhs_0 = 0, dual_feasibility_rhs_1 = 0, dual_feasibility_rhs_3 = 0;
  double primal_feasibility_lhs = 0, primal_feasibility_eq_lhs = 0, primal_feasibility_in_lhs = 0;
  double dual_feasibility_lhs = 0;
  double duality_gap = 0, rhs_duality_gap = 0;
  double scaled_eps = s.eps_abs;

  static const bool trace = std::getenv("PQO_TRACE") != nullptr; // debugging aid of the test infrastructure
  for (std::int64_t iter = 0; iter < s.max_iter; ++iter) {
    global_primal_residual(qp, primal_feasibility_lhs, primal_feasibility_eq_rhs_0,
                           primal_feasibility_in_rhs_0, primal_feasibility_eq_lhs,
                           primal_feasibility_in_lhs);
    global_dual_residual(qp, dual_feasibility_lhs, dual_feasibility_rhs_0, dual_feasibility_rhs_1,
                         dual_feasibility_rhs_3, rhs_duality_gap, duality_gap);
    r.info.pri_res = primal_feasibility_lhs;
    r.info.dua_res = dual_feasibility_lhs;
    r.info.duality_gap = duality_gap;

    double new_bcl_mu_in = r.info.mu_in;
    double new_bcl_mu_eq = r.info.mu_eq;
    double new_bcl_mu_in_inv = r.info.mu_in_inv;
    double new_bcl_mu_eq_inv = r.info.mu_eq_inv;

    double rhs_pri = scaled_eps;
    if (s.eps_rel != 0)
      rhs_pri += s.eps_rel * std::max(primal_feasibility_eq_rhs_0, primal_feasibility_in_rhs_0);
    bool is_primal_feasible = primal_feasibility_lhs <= rhs_pri;
    double rhs_dua = s.eps_abs;
    if (s.eps_rel != 0)
      rhs_dua += s.eps_rel * std::max(std::max(dual_feasibility_rhs_3, dual_feasibility_rhs_0),
                                      std::max(dual_feasibility_rhs_1, w.dual_feasibility_rhs_2));
    bool is_dual_feasible = dual_feasibility_lhs <= rhs_dua;

    // solver.hpp:1469-1510 — `verbose` is not only printing: the reference unscales x, y, z, evaluates the
    // objective, prints, and scales them back.  unscale followed by scale is the identity only up to
    // rounding, so a verbose run perturbs the iterates in their last bits at every outer iteration
    // (test/src/dense_qp_wrapper.cpp:7178 runs its closest-feasible family with verbose = true).
    if (s.verbose) {
      sc.unscale_primal(r.x.data());
      sc.unscale_dual_eq(r.y.data());
      sc.unscale_dual_in(r.z.data());
      if (qp.box_constraints)
        sc.unscale_box_dual_in(r.z.data() + n_in);
      r.info.objValue = objective_value(qp);
      if (qp.verbose_sink)
        qp.verbose_sink(iter + 1, r.info);
      {
        const double rec[8] = { 1.0, double(iter + 1), r.info.pri_res, r.info.dua_res, r.info.duality_gap, r.info.mu_in, r.info.rho, 0.0 };
        qp.trace.insert(qp.trace.end(), rec, rec + 8);
      }
      sc.scale_primal(r.x.data());
      sc.scale_dual_eq(r.y.data());
      sc.scale_dual_in(r.z.data());
      if (qp.box_constraints)
        sc.scale_box_dual_in(r.z.data() + n_in);
    }
    if (is_primal_feasible && is_dual_feasible) {
      if (s.check_duality_gap) {
        if (std::fabs(r.info.duality_gap) <= s.eps_duality_gap_abs + s.eps_duality_gap_rel * rhs_duality_gap) {
          if (s.primal_infeasibility_solving && r.info.status == PQP_PRIMAL_INFEASIBLE)
            r.info.status = PQP_SOLVED_CLOSEST_PRIMAL_FEASIBLE;
          else
            r.info.status = PQP_SOLVED;
          break;
        }
      } else {
        r.info.status = PQP_SOLVED;
        break;
      }
    }
    r.info.iter_ext += 1;
    w.x_prev = r.x;
    w.y_prev = r.y;
    w.z_prev = r.z;

    sc.scale_primal_residual_in(w.primal_residual_in_scaled_up.data());
    if (qp.box_constraints)
      sc.scale_box_primal_residual_in(w.primal_residual_in_scaled_up.data() + n_in);
    for (isize i = 0; i < nc; ++i)
      w.primal_residual_in_scaled_up[size_t(i)] += w.z_prev[size_t(i)] * r.info.mu_in;
    if (s.merit_function_type == PQP_MERIT_GPDAL)
      for (isize i = 0; i < nc; ++i)
        w.primal_residual_in_scaled_up[size_t(i)] += (s.alpha_gpdal - 1.) * r.info.mu_in * r.z[size_t(i)];
    r.si = w.primal_residual_in_scaled_up;
    for (isize i = 0; i < n_in; ++i) {
      w.primal_residual_in_scaled_up[size_t(i)] -= w.u_scaled[size_t(i)];
      r.si[size_t(i)] -= w.l_scaled[size_t(i)];
    }
    if (qp.box_constraints)
      for (isize i = 0; i < n; ++i) {
        w.primal_residual_in_scaled_up[size_t(n_in + i)] -= w.u_box_scaled[size_t(i)];
        r.si[size_t(n_in + i)] -= w.l_box_scaled[size_t(i)];
      }

    primal_dual_newton_semi_smooth(qp, bcl_eta_in);

    if ((r.info.status == PQP_PRIMAL_INFEASIBLE && !s.primal_infeasibility_solving) ||
        r.info.status == PQP_DUAL_INFEASIBLE) {
      for (isize k = 0; k < n; ++k)
        r.x[size_t(k)] = w.dw_aug[size_t(k)];
      for (isize k = 0; k < n_eq; ++k)
        r.y[size_t(k)] = w.dw_aug[size_t(n + k)];
      for (isize k = 0; k < nc; ++k)
        r.z[size_t(k)] = w.dw_aug[size_t(n + n_eq + k)];
      break;
    }
    if (scaled_eps == s.eps_abs && s.primal_infeasibility_solving &&
        r.info.status == PQP_PRIMAL_INFEASIBLE) {
      for (isize k = 0; k < n_eq + n_in; ++k)
        w.rhs[size_t(n + k)] = 1.0;
      for (isize k = 0; k < n; ++k)
        w.rhs[size_t(k)] = 0;
      for (isize i = 0; i < n_eq; ++i)
        for (isize k = 0; k < n; ++k)
          w.rhs[size_t(k)] += m.A[size_t(i * n + k)] * w.rhs[size_t(n + i)];
      for (isize i = 0; i < n_in; ++i)
        for (isize k = 0; k < n; ++k)
          w.rhs[size_t(k)] += m.C[size_t(i * n + k)] * w.rhs[size_t(n + n_eq + i)];
      if (qp.box_constraints)
        for (isize k = 0; k < n; ++k)
          w.rhs[size_t(k)] += w.i_scaled[size_t(k)];
      scaled_eps = infty_norm(w.rhs.data(), n) * s.eps_abs;
    }
    double primal_feasibility_lhs_new = primal_feasibility_lhs;
    global_primal_residual(qp, primal_feasibility_lhs_new, primal_feasibility_eq_rhs_0,
                           primal_feasibility_in_rhs_0, primal_feasibility_eq_lhs,
                           primal_feasibility_in_lhs);
    is_primal_feasible =
      primal_feasibility_lhs_new <=
      (scaled_eps + s.eps_rel * std::max(primal_feasibility_eq_rhs_0, primal_feasibility_in_rhs_0));
    r.info.pri_res = primal_feasibility_lhs_new;
    if (is_primal_feasible) {
      double dual_feasibility_lhs_new = dual_feasibility_lhs;
      global_dual_residual(qp, dual_feasibility_lhs_new, dual_feasibility_rhs_0,
                           dual_feasibility_rhs_1, dual_feasibility_rhs_3, rhs_duality_gap,
                           duality_gap);
      r.info.dua_res = dual_feasibility_lhs_new;
      r.info.duality_gap = duality_gap;
      is_dual_feasible =
        dual_feasibility_lhs_new <=
        (s.eps_abs + s.eps_rel * std::max(std::max(dual_feasibility_rhs_3, dual_feasibility_rhs_0),
                                          std::max(dual_feasibility_rhs_1, w.dual_feasibility_rhs_2)));
      if (is_dual_feasible) {
        bool gap_ok = true;
        if (s.check_duality_gap)
          gap_ok = std::fabs(r.info.duality_gap) <=
                   s.eps_duality_gap_abs + s.eps_duality_gap_rel * rhs_duality_gap;
        if (gap_ok) {
          if (s.primal_infeasibility_solving && r.info.status == PQP_PRIMAL_INFEASIBLE)
            r.info.status = PQP_SOLVED_CLOSEST_PRIMAL_FEASIBLE;
          else
            r.info.status = PQP_SOLVED;
        }
      }
    }
    if (s.bcl_update) {
      bcl_update(qp, primal_feasibility_lhs_new, bcl_eta_ext, bcl_eta_in, bcl_eta_ext_init,
                 eps_in_min, new_bcl_mu_in, new_bcl_mu_eq, new_bcl_mu_in_inv, new_bcl_mu_eq_inv);
    } else {
      Martinez_update(qp, primal_feasibility_lhs_new, primal_feasibility_lhs, bcl_eta_in, eps_in_min,
                      new_bcl_mu_in, new_bcl_mu_eq, new_bcl_mu_in_inv, new_bcl_mu_eq_inv);
    }
    double dual_feasibility_lhs_new = dual_feasibility_lhs;
    global_dual_residual(qp, dual_feasibility_lhs_new, dual_feasibility_rhs_0, dual_feasibility_rhs_1,
                         dual_feasibility_rhs_3, rhs_duality_gap, duality_gap);
    r.info.dua_res = dual_feasibility_lhs_new;
    r.info.duality_gap = duality_gap;
    if (primal_feasibility_lhs_new >= primal_feasibility_lhs &&
        dual_feasibility_lhs_new >= dual_feasibility_lhs && r.info.mu_in <= 1e-5) {
      new_bcl_mu_in = s.cold_reset_mu_in;
      new_bcl_mu_eq = s.cold_reset_mu_eq;
      new_bcl_mu_in_inv = s.cold_reset_mu_in_inv;
      new_bcl_mu_eq_inv = s.cold_reset_mu_eq_inv;
    }
    if (trace)
      std::fprintf(stderr, "[oracle] ext %lld status %d iter %lld n_c %lld mu_in %.3e->%.3e pri %.6e->%.6e dua %.6e->%.6e scaled_eps %.3e eta_ext %.3e eta_in %.3e\n",
                   (long long)iter, int(r.info.status), (long long)r.info.iter, (long long)w.n_c, r.info.mu_in, new_bcl_mu_in,
                   primal_feasibility_lhs, primal_feasibility_lhs_new, dual_feasibility_lhs, dual_feasibility_lhs_new, scaled_eps, bcl_eta_ext, bcl_eta_in);
    if (r.info.mu_in != new_bcl_mu_in || r.info.mu_eq != new_bcl_mu_eq) {
      ++r.info.mu_updates;
      mu_update(qp, new_bcl_mu_eq, new_bcl_mu_in);
    }
    r.info.mu_eq = new_bcl_mu_eq;
    r.info.mu_in = new_bcl_mu_in;
    r.info.mu_eq_inv = new_bcl_mu_eq_inv;
    r.info.mu_in_inv = new_bcl_mu_in_inv;
  }

  sc.unscale_primal(r.x.data());
  sc.unscale_dual_eq(r.y.data());
  sc.unscale_dual_in(r.z.data());
  if (qp.box_constraints)
    sc.unscale_box_dual_in(r.z.data() + n_in);
  if (s.primal_infeasibility_solving && r.info.status == PQP_PRIMAL_INFEASIBLE) {
    sc.unscale_primal_residual_eq(r.se.data());
    sc.unscale_primal_residual_in(r.si.data());
    if (qp.box_constraints)
      sc.unscale_box_primal_residual_in(r.si.data() + n_in);
  }
  r.info.objValue = objective_value(qp);
  if (s.compute_timings) {
    auto t1 = std::chrono::steady_clock::now();
    r.info.solve_time = std::chrono::duration<double, std::micro>(t1 - t0).count();
    r.info.run_time = r.info.solve_time + r.info.setup_time;
  }
  w.dirty = true;
  w.is_initialized = true;
}

// wrapper.hpp:81-113
int
dense_backend_choice(int backend, isize dim, isize n_eq, isize n_in, bool box)
{
  if (backend != PQP_BACKEND_AUTOMATIC)
    return backend;
  isize n_constraints = n_in + (box ? dim : 0);
  double threshold = 1.5, frequence = 0.2;
  double d = double(dim);
  double PrimalDualLDLTCost =
    0.5 * std::pow(double(n_eq) / d, 2) +
    0.17 * (std::pow(double(n_eq) / d, 3) + std::pow(double(n_constraints) / d, 3)) +
    frequence * std::pow(double(n_eq + n_constraints) / d, 2) / d;
  double PrimalLDLTCost = threshold * ((0.5 * double(n_eq) + double(n_constraints)) / d + frequence / d);
  return PrimalDualLDLTCost > PrimalLDLTCost ? PQP_BACKEND_PRIMAL_LDLT : PQP_BACKEND_PRIMAL_DUAL_LDLT;
}

} // namespace

// ------------------------------------------------------------------ wrapper.hpp
QP::QP(isize dim, isize n_eq, isize n_in, bool box, int hessian, int backend)
  : dense_backend(dense_backend_choice(backend, dim, n_eq, n_in, box))
  , box_constraints(box)
  , hessian_type(hessian)
{
  if (dim == 0)
    throw std::invalid_argument("wrong argument size: the dimension wrt the primal variable x "
                                "should be strictly positive."); // model.hpp:65-68
  const isize nc = n_in + (box ? dim : 0);
  pqp_settings_default(&settings, dense_backend);
  pqp_info_default(&results.info, dense_backend);
  results.x.assign(size_t(dim), 0.0);
  results.y.assign(size_t(n_eq), 0.0);
  results.z.assign(size_t(nc), 0.0);
  results.se.assign(size_t(n_eq), 0.0);
  results.si.assign(size_t(nc), 0.0);
  model.dim = dim;
  model.n_eq = n_eq;
  model.n_in = n_in;
  model.H.assign(size_t(dim * dim), 0.0);
  model.g.assign(size_t(dim), 0.0);
  model.A.assign(size_t(n_eq * dim), 0.0);
  model.C.assign(size_t(n_in * dim), 0.0);
  model.b.assign(size_t(n_eq), 0.0);
  model.u.assign(size_t(n_in), +infinite_bound());
  model.l.assign(size_t(n_in), -infinite_bound());
  if (box) {
    model.u_box.assign(size_t(dim), +infinite_bound());
    model.l_box.assign(size_t(dim), -infinite_bound());
  }
  Workspace& w = work;
  w.H_scaled.assign(size_t(dim * dim), 0.0);
  w.g_scaled.assign(size_t(dim), 0.0);
  w.A_scaled.assign(size_t(n_eq * dim), 0.0);
  w.C_scaled.assign(size_t(n_in * dim), 0.0);
  w.b_scaled.assign(size_t(n_eq), 0.0);
  w.u_scaled.assign(size_t(n_in), 0.0);
  w.l_scaled.assign(size_t(n_in), 0.0);
  if (box) {
    w.u_box_scaled.assign(size_t(dim), 0.0);
    w.l_box_scaled.assign(size_t(dim), 0.0);
    w.i_scaled.assign(size_t(dim), 1.0);
  }
  w.x_prev.assign(size_t(dim), 0.0);
  w.y_prev.assign(size_t(n_eq), 0.0);
  w.z_prev.assign(size_t(nc), 0.0);
  w.kkt.assign(size_t((dim + n_eq) * (dim + n_eq)), 0.0);
  w.ldl.reserve_uninit(dim + n_eq + nc);
  w.current_bijection_map.resize(size_t(nc));
  w.new_bijection_map.resize(size_t(nc));
  for (isize i = 0; i < nc; ++i) {
    w.current_bijection_map[size_t(i)] = i;
    w.new_bijection_map[size_t(i)] = i;
  }
  // NB: the reference leaves these uninitialised; false here.
  w.active_set_up.assign(size_t(nc), 0);
  w.active_set_low.assign(size_t(nc), 0);
  w.active_inequalities.assign(size_t(nc), 0);
  w.active_part_z.assign(size_t(nc), 0.0);
  w.dw_aug.assign(size_t(dim + n_eq + nc), 0.0);
  w.rhs.assign(size_t(dim + n_eq + nc), 0.0);
  w.err.assign(size_t(dim + n_eq + nc), 0.0);
  w.primal_residual_in_scaled_up.assign(size_t(nc), 0.0);
  w.primal_residual_in_scaled_up_plus_alphaCdx.assign(size_t(nc), 0.0);
  w.primal_residual_in_scaled_low_plus_alphaCdx.assign(size_t(nc), 0.0);
  w.Cdx.assign(size_t(nc), 0.0);
  w.alphas.reserve(size_t(2 * nc));
  w.Hdx.assign(size_t(dim), 0.0);
  w.Adx.assign(size_t(n_eq), 0.0);
  w.dual_residual_scaled.assign(size_t(dim), 0.0);
  w.CTz.assign(size_t(dim), 0.0);
  ruiz.delta.assign(size_t(dim + n_eq + nc), 1.0);
  ruiz.c = 1;
  ruiz.dim = dim;
  ruiz.n_eq = n_eq;
  ruiz.n_in = n_in;
}

void
QP::init(const double* H, const double* g, const double* A, const double* b, const double* C,
         const double* l, const double* u, const double* l_box, const double* u_box,
         bool compute_preconditioner, double rho, double mu_eq, double mu_in,
         double manual_minimal_H_eigenvalue)
{
  auto t0 = std::chrono::steady_clock::now();
  settings.compute_preconditioner = compute_preconditioner;
  if (settings.initial_guess == PQP_WARM_START_WITH_PREVIOUS_RESULT)
    work.refactorize = true;
  else
    work.refactorize = false;
  work.proximal_parameter_update = false;
  int preconditioner_status = compute_preconditioner ? 0 : 2;
  update_proximal_parameters(*this, rho, mu_eq, mu_in);
  update_default_rho_with_minimal_Hessian_eigen_value(*this, manual_minimal_H_eigenvalue);
  setup(*this, H, g, A, b, C, l, u, l_box, u_box, preconditioner_status);
  work.is_initialized = true;
  if (settings.compute_timings)
    results.info.setup_time =
      std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
}

void
QP::update(const double* H, const double* g, const double* A, const double* b, const double* C,
           const double* l, const double* u, const double* l_box, const double* u_box,
           bool update_preconditioner, double rho, double mu_eq, double mu_in,
           double manual_minimal_H_eigenvalue)
{
  settings.update_preconditioner = update_preconditioner;
  if (!work.is_initialized) {
    // wrapper.hpp:743-746 (NB: the eigenvalue estimate is not forwarded)
    init(H, g, A, b, C, l, u, l_box, u_box, update_preconditioner, rho, mu_eq, mu_in,
         std::numeric_limits<double>::quiet_NaN());
    return;
  }
  auto t0 = std::chrono::steady_clock::now();
  work.refactorize = false;
  work.proximal_parameter_update = false;
  int preconditioner_status = update_preconditioner ? 0 : 1;
  const isize n = model.dim, n_eq = model.n_eq, n_in = model.n_in;
  // helpers.hpp:372-480
  if (g)
    model.g.assign(g, g + n);
  if (b)
    model.b.assign(b, b + n_eq);
  if (u)
    model.u.assign(u, u + n_in);
  if (l)
    model.l.assign(l, l + n_in);
  if (u_box && box_constraints)
    model.u_box.assign(u_box, u_box + n);
  if (l_box && box_constraints)
    model.l_box.assign(l_box, l_box + n);
  if (H || A || C)
    work.refactorize = true;
  if (H)
    model.H.assign(H, H + n * n);
  if (A)
    model.A.assign(A, A + n_eq * n);
  if (C)
    model.C.assign(C, C + n_in * n);
  update_proximal_parameters(*this, rho, mu_eq, mu_in);
  update_default_rho_with_minimal_Hessian_eigen_value(*this, manual_minimal_H_eigenvalue);
  setup(*this, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr,
        preconditioner_status);
  if (settings.compute_timings)
    results.info.setup_time =
      std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
}

void
QP::solve(const double* x, const double* y, const double* z)
{
  // helpers.hpp:715-763
  if (x || y || z) {
    settings.initial_guess = PQP_WARM_START;
    if (x)
      results.x.assign(x, x + model.dim);
    if (y)
      results.y.assign(y, y + model.n_eq);
    if (z)
      results.z.assign(z, z + n_constraints());
  }
  qp_solve(*this);
}

void
QP::cleanup()
{
  results_cleanup(results, settings);
  work_cleanup(work, n_constraints());
}

// reference dense/compute_ECJ.hpp:29-132 (compute_backward) and :134-189
// (compute_backward_loss_ESG), restated literally -- including the way the inequality part of
// the right-hand side is scaled once per loop iteration (:100-112) and the index used for the
// inactive entries of dz (:139-146).  No box constraints (the reference ignores them here).
void
QP::compute_backward(const double* loss_derivative, double eps, double rho_new, double mu_new)
{
  if (results.info.status == PQP_DUAL_INFEASIBLE)
    throw std::invalid_argument("the QP problem is not feasible, so computing the derivatives is not valid "
                                "in this setting. Try enabling infeasible solving if the problem is only "
                                "primally infeasible.");
  const isize n = model.dim, n_eq = model.n_eq, n_in = model.n_in;
  Workspace& w = work;
  BackwardData& bd = backward_data;
  bd.dL_dH.assign(size_t(n * n), 0.0);
  bd.dL_dg.assign(size_t(n), 0.0);
  bd.dL_dA.assign(size_t(n_eq * n), 0.0);
  bd.dL_db.assign(size_t(n_eq), 0.0);
  bd.dL_dC.assign(size_t(n_in * n), 0.0);
  bd.dL_du.assign(size_t(n_in), 0.0);
  bd.dL_dl.assign(size_t(n_in), 0.0);
  // derive solution: active sets at (x, z) on the unscaled model (:48-57)
  isize numactive = 0;
  Vec ctz(size_t(n_in), 0.0); // the reference resizes work.CTz (dim entries) to n_in here
  for (isize i = 0; i < n_in; ++i) {
    double cx = 0;
    for (isize k = 0; k < n; ++k)
      cx += model.C[size_t(i * n + k)] * results.x[size_t(k)];
    ctz[size_t(i)] = cx + results.z[size_t(i)];
    w.active_set_up[size_t(i)] = (ctz[size_t(i)] - model.u[size_t(i)]) >= 0.;
    w.active_set_low[size_t(i)] = (ctz[size_t(i)] - model.l[size_t(i)]) <= 0.;
    w.active_inequalities[size_t(i)] = w.active_set_up[size_t(i)] || w.active_set_low[size_t(i)];
    numactive += w.active_inequalities[size_t(i)] ? 1 : 0;
  }
  const isize inner_pb_dim = n + n_eq + numactive;
  zero(w.rhs);
  results.info.rho = rho_new;
  results.info.mu_eq = mu_new;
  results.info.mu_in = mu_new;
  // factorisation from scratch with the new proximal parameters, then the active set (:66-86)
  setup_factorization(*this);
  w.n_c = 0;
  for (isize i = 0; i < n_in; ++i) {
    w.current_bijection_map[size_t(i)] = i;
    w.new_bijection_map[size_t(i)] = i;
  }
  active_set_change(*this);
  w.constraints_changed = false;
  // right-hand side (:88-112)
  for (isize i = 0; i < n + n_eq + n_in; ++i)
    w.rhs[size_t(i)] = -loss_derivative[i];
  for (isize k = 0; k < n; ++k)
    w.rhs[size_t(k)] *= ruiz.delta[size_t(k)] * ruiz.c; // scale_dual_residual_in_place
  bool eq_zero = true;
  for (isize k = 0; k < n_eq; ++k)
    eq_zero = eq_zero && w.rhs[size_t(n + k)] == 0.0;
  if (!eq_zero)
    for (isize k = 0; k < n_eq; ++k)
      w.rhs[size_t(n + k)] = -loss_derivative[n + k] * ruiz.delta[size_t(n + k)]; // ..._eq
  bool in_zero = true;
  for (isize k = 0; k < n_in; ++k)
    in_zero = in_zero && w.rhs[size_t(n + n_eq + k)] == 0.0;
  if (!in_zero) {
    for (isize i = 0; i < n_in; ++i) {
      const isize j = w.current_bijection_map[size_t(i)];
      if (j < w.n_c)
        w.rhs[size_t(j + n + n_eq)] = -loss_derivative[i + n + n_eq];
      for (isize k = 0; k < n_in; ++k) // scale_primal_residual_in_place_in, inside the loop (:107-111)
        w.rhs[size_t(n + n_eq + k)] *= ruiz.delta[size_t(n + n_eq + k)];
    }
  }
  iterative_solve_with_permut_fact(*this, eps, inner_pb_dim);
  // compute_backward_loss_ESG (:134-189)
  zero(w.active_part_z);
  for (isize j = 0; j < n_in; ++j) {
    const isize i = w.current_bijection_map[size_t(j)];
    if (i < w.n_c)
      w.active_part_z[size_t(j)] = w.dw_aug[size_t(n + n_eq + i)];
    else
      w.active_part_z[size_t(j)] = loss_derivative[n + n_eq + i];
  }
  for (isize j = 0; j < n_in; ++j)
    w.dw_aug[size_t(n + n_eq + j)] = w.active_part_z[size_t(j)];
  for (isize k = 0; k < n; ++k)
    w.dw_aug[size_t(k)] *= ruiz.delta[size_t(k)]; // unscale_primal_in_place
  for (isize k = 0; k < n_eq; ++k)
    w.dw_aug[size_t(n + k)] = w.dw_aug[size_t(n + k)] * ruiz.delta[size_t(n + k)] / ruiz.c;
  for (isize k = 0; k < n_in; ++k)
    w.dw_aug[size_t(n + n_eq + k)] =
      w.dw_aug[size_t(n + n_eq + k)] * ruiz.delta[size_t(n + n_eq + k)] / ruiz.c;
  const double* dx = w.dw_aug.data();
  const double* dy = w.dw_aug.data() + n;
  const double* dz = w.dw_aug.data() + n + n_eq;
  for (isize i = 0; i < n_in; ++i) {
    for (isize k = 0; k < n; ++k)
      bd.dL_dC[size_t(i * n + k)] = dz[i] * results.x[size_t(k)] + results.z[size_t(i)] * dx[k];
    bd.dL_du[size_t(i)] = w.active_set_up[size_t(i)] ? -dz[i] : 0.0;
    bd.dL_dl[size_t(i)] = w.active_set_low[size_t(i)] ? -dz[i] : 0.0;
  }
  for (isize i = 0; i < n_eq; ++i) {
    for (isize k = 0; k < n; ++k)
      bd.dL_dA[size_t(i * n + k)] = dy[i] * results.x[size_t(k)] + results.y[size_t(i)] * dx[k];
    bd.dL_db[size_t(i)] = -dy[i];
  }
  for (isize i = 0; i < n; ++i) {
    for (isize k = 0; k < n; ++k)
      bd.dL_dH[size_t(i * n + k)] = 0.5 * (dx[i] * results.x[size_t(k)] + results.x[size_t(i)] * dx[k]);
    bd.dL_dg[size_t(i)] = dx[i];
  }
}

} // namespace pqo
