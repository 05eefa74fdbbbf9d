// How many host cores does this box really give one process?  Pure-compute OpenMP scaling probe
// (no memory traffic, no allocation): each thread runs the same fixed FMA chain; the aggregate rate
// is printed for a sweep of team sizes.  Used to interpret bench.py's cpu_baseline thread sweep.
//   g++ -O2 -fopenmp scripts/microbench/omp_scaling.cpp -o /tmp/omp_scaling && /tmp/omp_scaling
#include <chrono>
#include <cstdio>
#include <initializer_list>
#include <omp.h>
#include <sched.h>
int main()
{
  cpu_set_t all;
  sched_getaffinity(0, sizeof(all), &all);
  const int procs = omp_get_num_procs();
  for (int pin : {0, 1})
    for (int nt : {1, 8, 16, 32, 64, 128, 256}) {
      if (nt > procs)
        continue;
      omp_set_num_threads(nt);
      double sink = 0;
      auto t0 = std::chrono::steady_clock::now();
#pragma omp parallel reduction(+ : sink)
      {
        if (pin) {
          cpu_set_t one;
          CPU_ZERO(&one);
          CPU_SET(omp_get_thread_num(), &one);
          sched_setaffinity(0, sizeof(one), &one);
        }
        double a = 1.0, b = 1.0, c = 1.0, d = 1.0;
        for (long i = 0; i < 400000000L; ++i) {
          a = a * 0.999999 + 1e-9;
          b = b * 0.999998 + 1e-9;
          c = c * 0.999997 + 1e-9;
          d = d * 0.999996 + 1e-9;
        }
        sink += a + b + c + d;
        if (pin && omp_get_thread_num() == 0)
          sched_setaffinity(0, sizeof(all), &all);
      }
      double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
      std::printf("pin=%d threads=%3d  time %.3f s  aggregate %.1f x single-thread work/s  (sink %g)\n", pin, nt, dt,
                  nt / dt, sink);
    }
}
