// TEST INFRASTRUCTURE ONLY -- never linked into, imported by or executed from
// the product path (proxsuite_amd/, include/proxsuite/, libproxqp_hip.so).
//
// CPU restatement (plain loops, no Eigen) of the reference's permuted dense
// LDL^T with low-rank modification:
//   reference include/proxsuite/linalg/dense/ldlt.hpp      (Ldlt<T>)
//   reference include/proxsuite/linalg/dense/factorize.hpp (factorization)
//   reference include/proxsuite/linalg/dense/update.hpp    (rank-r update)
//   reference include/proxsuite/linalg/dense/modify.hpp    (row/col insert, delete)
//   reference include/proxsuite/linalg/dense/solve.hpp     (solve)
// Each function cites the lines it follows.  Storage is column-major with the
// diagonal D stored on the diagonal of the unit-lower factor, as in the
// reference (ldlt.hpp:209-210, 616-629).
//
// Parity note: Eigen (the reference's substrate) is not available in the
// authoring container, so this restatement cannot be diffed against the
// reference binary; see oracle/README.md ("parity unpinned" for random QPs,
// pinned on the known-answer / fixture set).
#ifndef PQP_ORACLE_LDLT_HPP
#define PQP_ORACLE_LDLT_HPP

#include <algorithm>
#include <cassert>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

namespace pqo {

using isize = std::int64_t;

// Per-solve operation counters (F_alg / B_alg numerators).
struct OpCounters
{
  double fact_flops = 0;     // sum m_f^3/3
  double fact_bytes = 0;     // sum 8 m_f^2
  double level2_flops = 0;   // everything else (2 flops per FMA)
  void reset() { *this = OpCounters{}; }
};

struct Ldlt
{
  std::vector<double> ld; // column-major, leading dimension `stride`
  isize stride = 0;
  isize n = 0;
  std::vector<isize> perm;     // perm[internal position] = user index
  std::vector<isize> perm_inv; // perm_inv[user index] = internal position
  std::vector<double> maybe_sorted_diag;
  OpCounters* ctr = nullptr;

  double& at(isize i, isize j) { return ld[size_t(j * stride + i)]; }
  double at(isize i, isize j) const { return ld[size_t(j * stride + i)]; }
  isize dim() const { return n; }

  // ldlt.hpp:241-257
  void reserve_uninit(isize cap)
  {
    if (cap <= stride && cap * cap <= isize(ld.size()))
      return;
    stride = cap;
    ld.assign(size_t(cap * cap), 0.0);
    perm.reserve(size_t(cap));
    perm_inv.reserve(size_t(cap));
    maybe_sorted_diag.reserve(size_t(cap));
  }

  // factorize.hpp:89-148 (left-looking, unblocked)
  static void factorize_unblocked(double* m, isize n, isize s, double* work)
  {
    if (n == 0)
      return;
    isize j = 0;
    while (true) {
      // work = l10^T .* d0 ; d_j -= work . l10
      double acc = 0;
      for (isize k = 0; k < j; ++k) {
        work[k] = m[k * s + j] * m[k * s + k];
        acc += work[k] * m[k * s + j];
      }
      m[j * s + j] -= acc;
      if (j + 1 == n)
        break;
      isize rem = n - j - 1;
      // l21 -= l20 * work
      double* l21 = m + j * s + j + 1;
      for (isize k = 0; k < j; ++k) {
        const double* l20k = m + k * s + j + 1;
        double wk = work[k];
        for (isize i = 0; i < rem; ++i)
          l21[i] -= l20k[i] * wk;
      }
      double inv = 1 / m[j * s + j];
      for (isize i = 0; i < rem; ++i)
        l21[i] *= inv;
      ++j;
    }
  }

  // X <- X * L^{-T} with L unit lower (n x n), X is rows x n; both col-major.
  // Restates `trans(l).triangularView<UnitUpper>().solveInPlace<OnTheRight>(x)`
  // (factorize.hpp:198-200, 254-256; modify.hpp:200-202, 238-240).
  static void trsm_right_unit_lower_t(const double* l,
                                      isize ls,
                                      isize n,
                                      double* x,
                                      isize xs,
                                      isize rows)
  {
    for (isize k = 0; k < n; ++k) {
      double* xk = x + k * xs;
      for (isize j = 0; j < k; ++j) {
        double lkj = l[j * ls + k];
        if (lkj == 0)
          continue;
        const double* xj = x + j * xs;
        for (isize i = 0; i < rows; ++i)
          xk[i] -= xj[i] * lkj;
      }
    }
  }

  // factorize.hpp:215-280 (recursive right-looking, leaf < 32 unblocked)
  static void factorize_recursive(double* m, isize n, isize s, std::vector<double>& scratch)
  {
    if (n < 32) {
      if (isize(scratch.size()) < n)
        scratch.resize(size_t(n));
      factorize_unblocked(m, n, s, scratch.data());
      return;
    }
    isize bs = (n + 1) / 2;
    isize rem = n - bs;
    double* l00 = m;
    double* l10 = m + bs;
    double* l11 = m + bs * s + bs;
    factorize_recursive(l00, bs, s, scratch);
    trsm_right_unit_lower_t(l00, s, bs, l10, s, rem);
    {
      std::vector<double> work(size_t(rem * bs));
      for (isize k = 0; k < bs; ++k) {
        double inv = 1 / l00[k * s + k];
        for (isize i = 0; i < rem; ++i) {
          work[size_t(k * rem + i)] = l10[k * s + i];
          l10[k * s + i] *= inv;
        }
      }
      // l11.lower -= l10 * work^T
      for (isize c = 0; c < rem; ++c) {
        double* col = l11 + c * s;
        for (isize k = 0; k < bs; ++k) {
          double w = work[size_t(k * rem + c)];
          if (w == 0)
            continue;
          const double* lk = l10 + k * s;
          for (isize i = c; i < rem; ++i)
            col[i] -= lk[i] * w;
        }
      }
    }
    factorize_recursive(l11, rem, s, scratch);
  }

  // ldlt.hpp:718-744 ; permutation factorize.hpp:17-46, 62-87.
  // `sym(i,j)` returns the (symmetric) matrix entry; only i>=j is requested.
  template<typename Sym>
  void factorize(isize n_, Sym sym)
  {
    reserve_uninit(n_);
    n = n_;
    perm.resize(size_t(n));
    perm_inv.resize(size_t(n));
    maybe_sorted_diag.resize(size_t(n));
    for (isize k = 0; k < n; ++k)
      perm[size_t(k)] = k;
    std::sort(perm.begin(), perm.end(), [&](isize i, isize j) {
      double lhs = std::fabs(sym(i, i));
      double rhs = std::fabs(sym(j, j));
      if (lhs == rhs)
        return i < j;
      return lhs > rhs;
    });
    for (isize k = 0; k < n; ++k)
      perm_inv[size_t(perm[size_t(k)])] = k;
    for (isize j = 0; j < n; ++j)
      for (isize i = j; i < n; ++i) {
        isize pi = perm[size_t(i)], pj = perm[size_t(j)];
        at(i, j) = pi >= pj ? sym(pi, pj) : sym(pj, pi);
      }
    for (isize i = 0; i < n; ++i)
      maybe_sorted_diag[size_t(i)] = at(i, i);
    std::vector<double> scratch;
    factorize_recursive(ld.data(), n, stride, scratch);
    if (ctr) {
      ctr->fact_flops += double(n) * double(n) * double(n) / 3.0;
      ctr->fact_bytes += 8.0 * double(n) * double(n);
    }
  }

  // solve.hpp:15-26 + ldlt.hpp:767-782
  void solve_in_place(double* rhs, isize m, std::vector<double>& work) const
  {
    assert(m == n);
    if (isize(work.size()) < m)
      work.resize(size_t(m));
    for (isize i = 0; i < m; ++i)
      work[size_t(i)] = rhs[perm[size_t(i)]];
    // forward, unit lower (column-oriented)
    for (isize j = 0; j < m; ++j) {
      double xj = work[size_t(j)];
      const double* col = ld.data() + j * stride;
      for (isize i = j + 1; i < m; ++i)
        work[size_t(i)] -= col[i] * xj;
    }
    for (isize j = 0; j < m; ++j)
      work[size_t(j)] /= at(j, j);
    // backward, unit upper = L^T (dot-product oriented over the column)
    for (isize j = m - 1; j >= 0; --j) {
      const double* col = ld.data() + j * stride;
      double acc = work[size_t(j)];
      for (isize i = j + 1; i < m; ++i)
        acc -= col[i] * work[size_t(i)];
      work[size_t(j)] = acc;
    }
    for (isize i = 0; i < m; ++i)
      rhs[i] = work[size_t(perm_inv[size_t(i)])];
    if (ctr)
      ctr->level2_flops += 2.0 * double(m) * double(m);
  }

  template<int K>
  static void rank_chunk_columns(double* __restrict__ inout_l, double* __restrict__ w0, isize w_stride, isize rem,
                                 const double* p_array, const double* mu_array)
  {
    double p[K], mu[K];
    for (int k = 0; k < K; ++k) {
      p[k] = p_array[k];
      mu[k] = mu_array[k];
    }
#pragma omp simd
    for (isize i = 0; i < rem; ++i) {
      double in_l = inout_l[i];
      for (int k = 0; k < K; ++k) {
        double wr = w0[k * w_stride + i];
        wr = std::fma(-p[k], in_l, wr);
        in_l = std::fma(mu[k], wr, in_l);
        w0[k * w_stride + i] = wr;
      }
      inout_l[i] = in_l;
    }
  }

  // update.hpp:219-287.  `r_fn()` is called once per column and returns how
  // many of the r updates are active from this column on.
  template<typename RFn>
  void rank_r_update_clobber_w_impl(double* l,
                                    isize ln,
                                    double* pw,
                                    isize w_stride,
                                    double* palpha,
                                    RFn r_fn)
  {
    for (isize j = 0; j < ln; ++j) {
      isize r = r_fn();
      isize r_done = 0;
      if (!(r_done < r))
        continue;
      while (true) {
        isize r_chunk = std::min<isize>(4, r - r_done);
        double p_array[4];
        double mu_array[4];
        double dj = l[j * stride + j];
        for (isize k = 0; k < r_chunk; ++k) {
          double& alpha = palpha[r_done + k];
          double p = pw[(r_done + k) * w_stride];
          double new_dj = dj + (alpha * p) * p;
          double mu = (alpha * p) / new_dj;
          alpha -= new_dj * (mu * mu);
          dj = new_dj;
          p_array[k] = p;
          mu_array[k] = mu;
        }
        l[j * stride + j] = dj;
        isize rem = ln - j - 1;
        double* inout_l = l + j * stride + j + 1;
        double* w0 = pw + 1 + r_done * w_stride;
        // (one instance per chunk width, the column loop vectorised over i -- the reference does the same with explicit
        // SIMD packs, update.hpp:155-204; every element still sees the same chain of fused multiply-adds: same bits)
        switch (r_chunk) {
          case 1:
            rank_chunk_columns<1>(inout_l, w0, w_stride, rem, p_array, mu_array);
            break;
          case 2:
            rank_chunk_columns<2>(inout_l, w0, w_stride, rem, p_array, mu_array);
            break;
          case 3:
            rank_chunk_columns<3>(inout_l, w0, w_stride, rem, p_array, mu_array);
            break;
          default:
            rank_chunk_columns<4>(inout_l, w0, w_stride, rem, p_array, mu_array);
            break;
        }
        if (ctr)
          ctr->level2_flops += 4.0 * double(rem) * double(r_chunk);
        r_done += r_chunk;
        if (!(r_done < r))
          break;
      }
      ++pw;
    }
  }

  // modify.hpp:58-79
  struct IndicesR
  {
    isize current_col;
    isize current_r;
    isize r;
    const isize* indices;
    isize operator()()
    {
      if (current_r == r)
        return current_r;
      while (current_col == indices[current_r] - current_r) {
        ++current_r;
        if (current_r == r)
          return current_r;
      }
      ++current_col;
      return current_r;
    }
  };
  struct ConstantR
  {
    isize r;
    isize operator()() const { return r; }
  };

  // ldlt.hpp:340-387 + modify.hpp:19-46, 80-127
  void delete_at(const isize* indices, isize r)
  {
    if (r == 0)
      return;
    isize nn = n;
    std::vector<isize> ia(static_cast<size_t>(r));
    for (isize k = 0; k < r; ++k)
      ia[size_t(k)] = perm_inv[size_t(indices[k])];
    std::sort(ia.begin(), ia.end());
    isize first = ia[0];
    isize w_stride = nn - first - r;
    if (w_stride < 1)
      w_stride = 1;
    std::vector<double> w(size_t(r * w_stride), 0.0);
    std::vector<double> alpha(static_cast<size_t>(r));
    for (isize k = 0; k < r; ++k) {
      isize j = ia[size_t(k)];
      alpha[size_t(k)] = at(j, j);
      double* pwk = w.data() + k * w_stride;
      for (isize chunk_i = k + 1; chunk_i < r + 1; ++chunk_i) {
        isize i_start = ia[size_t(chunk_i - 1)] + 1;
        isize i_finish = chunk_i == r ? nn : ia[size_t(chunk_i)];
        for (isize i = i_start; i < i_finish; ++i)
          pwk[i - chunk_i - first] = at(i, j);
      }
    }
    // compaction (modify.hpp:19-46)
    for (isize chunk_j = 0; chunk_j < r + 1; ++chunk_j) {
      isize j_start = chunk_j == 0 ? 0 : ia[size_t(chunk_j - 1)] + 1;
      isize j_finish = chunk_j == r ? nn : ia[size_t(chunk_j)];
      for (isize j = j_start; j < j_finish; ++j) {
        for (isize chunk_i = chunk_j; chunk_i < r + 1; ++chunk_i) {
          isize i_start = chunk_i == chunk_j ? j : ia[size_t(chunk_i - 1)] + 1;
          isize i_finish = chunk_i == r ? nn : ia[size_t(chunk_i)];
          if (chunk_i != 0 || chunk_j != 0) {
            for (isize i = i_start; i < i_finish; ++i)
              at(i - chunk_i, j - chunk_j) = at(i, j);
          }
        }
      }
    }
    isize sub = nn - first - r;
    rank_r_update_clobber_w_impl(ld.data() + first * stride + first,
                                 sub,
                                 w.data(),
                                 w_stride,
                                 alpha.data(),
                                 IndicesR{ first, 0, r, ia.data() });
    // permutation bookkeeping (ldlt.hpp:367-386)
    for (isize k = 0; k < r; ++k) {
      isize i_actual = ia[size_t(r - 1 - k)];
      isize i = indices[r - 1 - k];
      perm.erase(perm.begin() + i_actual);
      perm_inv.erase(perm_inv.begin() + i);
      maybe_sorted_diag.erase(maybe_sorted_diag.begin() + i_actual);
      for (isize j = 0; j < nn - 1 - k; ++j) {
        if (perm[size_t(j)] > i)
          --perm[size_t(j)];
        if (perm_inv[size_t(j)] > i_actual)
          --perm_inv[size_t(j)];
      }
    }
    n = nn - r;
  }

  // ldlt.hpp:389-401
  isize choose_insertion_position(double diag_elem) const
  {
    isize pos = 0;
    for (; pos < n; ++pos)
      if (diag_elem >= maybe_sorted_diag[size_t(pos)])
        break;
    return pos;
  }

  // ldlt.hpp:431-475 + modify.hpp:129-264.  `a` is (n+r) x r column-major with
  // leading dimension `as`, holding the new columns in *user* index order.
  void insert_block_at(isize i, const double* a, isize as, isize r)
  {
    if (r == 0)
      return;
    isize old_n = n;
    isize new_n = n + r;
    assert(new_n <= stride);
    isize pos = choose_insertion_position(a[0 * as + i]);
    for (isize j = 0; j < old_n; ++j) {
      if (perm[size_t(j)] >= i)
        perm[size_t(j)] += r;
      if (perm_inv[size_t(j)] >= pos)
        perm_inv[size_t(j)] += r;
    }
    for (isize k = 0; k < r; ++k) {
      perm.insert(perm.begin() + (pos + k), i + k);
      perm_inv.insert(perm_inv.begin() + (i + k), pos + k);
      maybe_sorted_diag.insert(maybe_sorted_diag.begin() + (pos + k), a[k * as + i + k]);
    }
    n = new_n;
    std::vector<double> pa(size_t(new_n * r));
    for (isize k = 0; k < r; ++k)
      for (isize j = 0; j < new_n; ++j)
        pa[size_t(k * new_n + j)] = a[k * as + perm[size_t(j)]];

    // shift storage (modify.hpp:145-179)
    isize current_col = old_n;
    while (current_col != pos) {
      --current_col;
      double* src = ld.data() + current_col * stride;
      double* dst = ld.data() + (current_col + r) * stride;
      std::move_backward(src + pos, src + old_n, dst + new_n);
      std::move_backward(src, src + pos, dst + pos);
    }
    while (current_col != 0) {
      --current_col;
      double* src = ld.data() + current_col * stride;
      std::move_backward(src + pos, src + old_n, src + new_n);
    }
    isize rem = new_n - pos - r;
    double* base = ld.data();
    double* l10 = base + pos;                       // r x pos
    double* l20 = base + pos + r;                   // rem x pos
    double* ld11 = base + pos * stride + pos;       // r x r
    double* l21 = base + pos * stride + pos + r;    // rem x r
    double* ld22 = base + (pos + r) * stride + pos + r;
    const double* a01 = pa.data();       // rows [0,pos)
    const double* a11 = pa.data() + pos; // rows [pos,pos+r)
    const double* a21 = pa.data() + pos + r;

    if (pos > 0) {
      for (isize c = 0; c < pos; ++c)
        for (isize k = 0; k < r; ++k)
          l10[c * stride + k] = a01[k * new_n + c];
      trsm_right_unit_lower_t(base, stride, pos, l10, stride, r);
      for (isize c = 0; c < pos; ++c) {
        double inv = 1 / base[c * stride + c];
        for (isize k = 0; k < r; ++k)
          l10[c * stride + k] *= inv;
      }
    }
    std::vector<double> d0xl10T(size_t(std::max<isize>(pos, 1) * r));
    for (isize k = 0; k < r; ++k)
      for (isize ii = k; ii < r; ++ii)
        ld11[k * stride + ii] = a11[k * new_n + ii];
    if (pos > 0) {
      for (isize k = 0; k < r; ++k)
        for (isize c = 0; c < pos; ++c)
          d0xl10T[size_t(k * pos + c)] = base[c * stride + c] * l10[c * stride + k];
      for (isize k = 0; k < r; ++k)
        for (isize ii = k; ii < r; ++ii) {
          double acc = 0;
          for (isize c = 0; c < pos; ++c)
            acc += l10[c * stride + ii] * d0xl10T[size_t(k * pos + c)];
          ld11[k * stride + ii] -= acc;
        }
    }
    for (isize k = 0; k < r; ++k) {
      double* l21k = l21 + k * stride;
      for (isize ii = 0; ii < rem; ++ii)
        l21k[ii] = a21[k * new_n + ii];
      for (isize c = 0; c < pos; ++c) {
        double wv = d0xl10T[size_t(k * pos + c)];
        if (wv == 0)
          continue;
        const double* l20c = l20 + c * stride;
        for (isize ii = 0; ii < rem; ++ii)
          l21k[ii] -= l20c[ii] * wv;
      }
    }
    if (ctr)
      ctr->level2_flops += 2.0 * double(r) * double(pos) * double(pos + rem);
    {
      std::vector<double> scratch;
      factorize_recursive(ld11, r, stride, scratch);
    }
    trsm_right_unit_lower_t(ld11, stride, r, l21, stride, rem);
    for (isize k = 0; k < r; ++k) {
      double inv = 1 / ld11[k * stride + k];
      for (isize ii = 0; ii < rem; ++ii)
        l21[k * stride + ii] *= inv;
    }
    if (rem > 0) {
      isize w_stride = rem;
      std::vector<double> w(size_t(r * w_stride));
      std::vector<double> alpha(static_cast<size_t>(r));
      for (isize k = 0; k < r; ++k) {
        alpha[size_t(k)] = -ld11[k * stride + k];
        std::copy(l21 + k * stride, l21 + k * stride + rem, w.data() + k * w_stride);
      }
      rank_r_update_clobber_w_impl(ld22, rem, w.data(), w_stride, alpha.data(), ConstantR{ r });
    }
  }

  // ldlt.hpp:516-570
  void diagonal_update_clobber_indices(isize* indices, isize r, const double* alpha_in)
  {
    if (r == 0)
      return;
    std::vector<isize> positions(static_cast<size_t>(r));
    std::vector<isize> sorted_indices(static_cast<size_t>(r));
    for (isize k = 0; k < r; ++k) {
      indices[k] = perm_inv[size_t(indices[k])];
      positions[size_t(k)] = k;
    }
    std::sort(positions.begin(), positions.end(), [&](isize i, isize j) {
      return indices[i] < indices[j];
    });
    for (isize k = 0; k < r; ++k)
      sorted_indices[size_t(k)] = indices[positions[size_t(k)]];
    isize first = sorted_indices[0];
    isize sub = n - first;
    std::vector<double> w(size_t(sub * r), 0.0);
    std::vector<double> alpha(static_cast<size_t>(r));
    for (isize k = 0; k < r; ++k) {
      alpha[size_t(k)] = alpha_in[positions[size_t(k)]];
      w[size_t(k * sub + sorted_indices[size_t(k)] - first)] = 1;
    }
    rank_r_update_clobber_w_impl(ld.data() + first * stride + first,
                                 sub,
                                 w.data(),
                                 sub,
                                 alpha.data(),
                                 IndicesR{ first, 0, r, sorted_indices.data() });
  }

  // debugging helper: P^T L D L^T P in user index order (ldlt.hpp:812-827)
  std::vector<double> reconstructed_matrix() const
  {
    std::vector<double> tmp(size_t(n * n), 0.0), out(size_t(n * n), 0.0);
    for (isize i = 0; i < n; ++i)
      for (isize j = 0; j < n; ++j) {
        double acc = 0;
        for (isize k = 0; k <= std::min(i, j); ++k) {
          double lik = (i == k) ? 1.0 : at(i, k);
          double ljk = (j == k) ? 1.0 : at(j, k);
          acc += lik * at(k, k) * ljk;
        }
        tmp[size_t(i * n + j)] = acc;
      }
    for (isize i = 0; i < n; ++i)
      for (isize j = 0; j < n; ++j)
        out[size_t(i * n + j)] = tmp[size_t(perm_inv[size_t(i)] * n + perm_inv[size_t(j)])];
    return out;
  }
};

} // namespace pqo

#endif
