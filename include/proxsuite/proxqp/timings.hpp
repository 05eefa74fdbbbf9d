// proxsuite/proxqp/timings.hpp -- the stopwatch of the reference's public headers (include/proxsuite/proxqp/timings.hpp:
// `Timer<T>` with start / stop / resume / elapsed, `elapsed().user` in microseconds), which its benchmark programs and
// user code time their loops with (benchmark/timings-*.cpp).  The reference pulls it in through dense/workspace.hpp; here
// dense/dense.hpp includes it.  Wall-clock time on std::chrono::steady_clock, reported in the `user` field as there.
#ifndef PROXSUITE_AMD_PROXQP_TIMINGS_HPP
#define PROXSUITE_AMD_PROXQP_TIMINGS_HPP

#include <chrono>

namespace proxsuite {
namespace proxqp {

struct CPUTimes
{
  double wall = 0;
  double user = 0; // microseconds
  double system = 0;
  void clear() { wall = user = system = 0; }
};

template<typename T>
struct Timer
{
  using clock = std::chrono::steady_clock;
  Timer() { start(); } // (a new stopwatch runs)

  // microseconds accumulated so far (the running lap included)
  CPUTimes elapsed() const
  {
    CPUTimes t = acc_;
    if (running_)
      t.user += lap_us(clock::now());
    return t;
  }
  // a stopped watch is reset and started; a running one keeps running
  void start()
  {
    if (running_)
      return;
    acc_.clear();
    running_ = true;
    t0_ = clock::now();
  }
  void stop()
  {
    if (!running_)
      return;
    acc_.user += lap_us(clock::now());
    running_ = false;
  }
  // continue a stopped watch without resetting it
  void resume()
  {
    if (running_)
      return;
    running_ = true;
    t0_ = clock::now();
  }
  bool is_stopped() const { return !running_; }

private:
  double lap_us(clock::time_point now) const
  {
    return double(std::chrono::duration_cast<std::chrono::nanoseconds>(now - t0_).count()) * 1e-3;
  }
  CPUTimes acc_;
  bool running_ = false;
  clock::time_point t0_;
};

} // namespace proxqp
} // namespace proxsuite

#endif
