// TEST INFRASTRUCTURE ONLY -- driver of the REAL ProxSuite dense backend (headers included in place
// from /root/reference/include; nothing of the reference is copied into this repository).
//
// Built by oracle/ref/build_ref.sh into oracle/_ref/ref_batchqp whenever Eigen3 is installed (the
// reference is header-only on top of Eigen, which /root/reference does not vendor).  Two uses:
//   1. golden vectors: `ref_batchqp golden <B> <n> <n_eq> <n_in> <out.bin>` solves the QPs of the
//      reference benchmark's generator (benchmark/timings-parallel.cpp:43-63: seed i for QP i,
//      dense_strongly_convex_qp(n, n_eq, n_in, 0.15, 1e-2), eps_abs 1e-9, eps_rel 0,
//      NO_INITIAL_GUESS) with dense::solve_in_parallel(BatchQP) and dumps (x, y, z, iter, iter_ext,
//      status, pri_res, dua_res) per QP -- tests/golden/make_reference_fixtures.py turns that into
//      the .npz fixtures the parity tests load;
//   2. CPU baseline of bench.py with kind "reference": `ref_batchqp time <B> <n> <n_eq> <n_in>
//      <passes> <threads>` times back-to-back solve_in_parallel passes over an init-ed BatchQP
//      exactly like benchmark/timings-parallel.cpp:211-220 and prints one JSON line.
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include <proxsuite/proxqp/dense/dense.hpp>
#include <proxsuite/proxqp/parallel/qp_solve.hpp>
#include <proxsuite/proxqp/utils/random_qp_problems.hpp>

using T = double;
using namespace proxsuite;
using namespace proxsuite::proxqp;

static void
fill(dense::BatchQP<T>& qps, int B, dense::isize n, dense::isize ne, dense::isize ni)
{
  for (int i = 0; i < B; ++i) {
    utils::rand::set_seed(std::uint64_t(i));
    dense::Model<T> m = utils::dense_strongly_convex_qp(n, ne, ni, 0.15, T(1e-2));
    auto& qp = qps.init_qp_in_place(n, ne, ni);
    qp.settings.eps_abs = T(1e-9);
    qp.settings.eps_rel = 0;
    qp.settings.initial_guess = InitialGuessStatus::NO_INITIAL_GUESS;
    qp.init(m.H, m.g, m.A, m.b, m.C, m.l, m.u);
  }
}

int
main(int argc, char** argv)
{
  if (argc < 6) {
    std::fprintf(stderr, "usage: %s golden B n n_eq n_in out.bin | time B n n_eq n_in passes threads\n", argv[0]);
    return 2;
  }
  const std::string mode = argv[1];
  const int B = std::atoi(argv[2]);
  const dense::isize n = std::atoi(argv[3]), ne = std::atoi(argv[4]), ni = std::atoi(argv[5]);
  dense::BatchQP<T> qps(size_t(B));
  fill(qps, B, n, ne, ni);
  if (mode == "golden") {
    if (argc < 7)
      return 2;
    dense::solve_in_parallel(qps);
    FILE* f = std::fopen(argv[6], "wb");
    if (!f)
      return 3;
    const std::int64_t hdr[4] = { B, std::int64_t(n), std::int64_t(ne), std::int64_t(ni) };
    std::fwrite(hdr, sizeof(hdr), 1, f);
    for (int i = 0; i < B; ++i) {
      auto& r = qps[dense::isize(i)].results;
      std::fwrite(r.x.data(), sizeof(T), size_t(n), f);
      std::fwrite(r.y.data(), sizeof(T), size_t(ne), f);
      std::fwrite(r.z.data(), sizeof(T), size_t(ni), f);
      const double meta[5] = { double(r.info.iter), double(r.info.iter_ext), double(int(r.info.status)),
                               double(r.info.pri_res), double(r.info.dua_res) };
      std::fwrite(meta, sizeof(meta), 1, f);
    }
    std::fclose(f);
    return 0;
  }
  if (mode == "time") {
    const int passes = argc > 6 ? std::atoi(argv[6]) : 10;
    const size_t threads = argc > 7 ? size_t(std::atoi(argv[7])) : 0;
    auto run = [&]() {
      if (threads)
        dense::solve_in_parallel(qps, optional<size_t>(threads));
      else
        dense::solve_in_parallel(qps);
    };
    run(); // first solve; the timed ones take the dirty path like the reference benchmark's
    const auto t0 = std::chrono::steady_clock::now();
    for (int p = 0; p < passes; ++p)
      run();
    const double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    int unsolved = 0;
    for (int i = 0; i < B; ++i)
      unsolved += qps[dense::isize(i)].results.info.status != QPSolverOutput::PROXQP_SOLVED;
    std::printf("{\"qps_per_s\": %.6f, \"passes\": %d, \"batch\": %d, \"threads\": %zu, \"unsolved\": %d}\n",
                double(B) * passes / s, passes, B, threads, unsolved);
    return 0;
  }
  return 2;
}
