// TEST INFRASTRUCTURE ONLY -- fiber scheduler of the SIMT emulator (see hip_emu.hpp).
#include "hip_emu.hpp"

#include <atomic>
#include <chrono>
#include <thread>

#undef threadIdx
#undef blockIdx
#undef blockDim
#undef gridDim

extern "C" void
hipemu_switch(void** save_sp, void* load_sp);

// x86-64 SysV cooperative context switch: callee-saved registers only.
asm(R"(
.text
.globl hipemu_switch
.type hipemu_switch,@function
hipemu_switch:
  pushq %rbp
  pushq %rbx
  pushq %r12
  pushq %r13
  pushq %r14
  pushq %r15
  movq %rsp, (%rdi)
  movq %rsi, %rsp
  popq %r15
  popq %r14
  popq %r13
  popq %r12
  popq %rbx
  popq %rbp
  ret
.size hipemu_switch,.-hipemu_switch
)");

namespace hipemu {

struct Fiber
{
  void* sp = nullptr;
  char* stack = nullptr;
  bool done = false;
  unsigned tid = 0;
};

thread_local Block* g_block = nullptr;
thread_local uint3_emu g_threadIdx, g_blockIdx;
thread_local dim3 g_blockDim, g_gridDim;

namespace {
constexpr size_t STACK_BYTES = 256 * 1024;
thread_local void* t_main_sp = nullptr;
thread_local const std::function<void()>* t_body = nullptr;
thread_local unsigned long t_idle_spins = 0;
thread_local std::vector<char*>* t_stack_pool = nullptr;

void
progress()
{
  t_idle_spins = 0;
}

void
switch_to(Block* b, unsigned from, int to)
{
  Fiber* f = b->fibers[from];
  if (to < 0) {
    hipemu_switch(&f->sp, t_main_sp);
  } else {
    b->cur = unsigned(to);
    hipemu_switch(&f->sp, b->fibers[size_t(to)]->sp);
  }
  // resumed
  g_threadIdx.x = f->tid;
  g_threadIdx.y = 0;
  g_threadIdx.z = 0;
}

// HIPEMU_ORDER=reverse schedules the fibers of a block in descending thread order; running
// the tests in both orders exposes most missing barriers (a value read before the thread
// that produces it has run).
bool
reverse_order()
{
  static const bool rev = [] {
    const char* e = std::getenv("HIPEMU_ORDER");
    return e && std::strcmp(e, "reverse") == 0;
  }();
  return rev;
}

int
next_alive(Block* b, unsigned from)
{
  const unsigned n = b->nthreads;
  for (unsigned k = 1; k <= n; ++k) {
    unsigned c = reverse_order() ? (from + n - k) % n : (from + k) % n;
    if (!b->fibers[c]->done)
      return int(c);
  }
  return -1;
}

void
fiber_exit()
{
  Block* b = g_block;
  unsigned me = b->cur;
  Fiber* f = b->fibers[me];
  f->done = true;
  b->alive--;
  Block::Wave& w = b->waves[me / 64];
  w.alive--;
  progress();
  // an exited thread no longer participates in barriers
  if (b->alive > 0 && b->bar_count >= b->alive && b->bar_count > 0) {
    b->bar_count = 0;
    b->bar_gen++;
  }
  if (w.alive > 0 && w.count >= w.alive && w.count > 0) {
    w.count = 0;
    w.gen++;
  }
  int nx = next_alive(b, me);
  switch_to(b, me, nx);
  fatal("resumed a finished fiber");
}

extern "C" void
hipemu_trampoline()
{
  Block* b = g_block;
  g_threadIdx.x = b->fibers[b->cur]->tid;
  g_threadIdx.y = 0;
  g_threadIdx.z = 0;
  (*t_body)();
  fiber_exit();
}

void
prepare_fiber(Fiber* f, unsigned tid, char* stack)
{
  f->tid = tid;
  f->done = false;
  f->stack = stack;
  uintptr_t top = reinterpret_cast<uintptr_t>(stack + STACK_BYTES);
  top &= ~uintptr_t(15);
  uintptr_t ret_slot = top - 16; // 16-aligned; after `ret`, rsp == ret_slot+8 (== 8 mod 16)
  *reinterpret_cast<void**>(ret_slot) = reinterpret_cast<void*>(&hipemu_trampoline);
  uintptr_t sp = ret_slot - 6 * 8;
  std::memset(reinterpret_cast<void*>(sp), 0, 6 * 8);
  f->sp = reinterpret_cast<void*>(sp);
}

void
run_block(unsigned bx, dim3 grid, dim3 block, size_t shmem, const std::function<void()>& body)
{
  Block b;
  b.nthreads = block.x;
  b.alive = block.x;
  b.waves.resize((block.x + 63) / 64);
  for (unsigned w = 0; w < b.waves.size(); ++w) {
    unsigned lo = w * 64, hi = std::min(block.x, lo + 64);
    b.waves[w].alive = hi - lo;
  }
  // LDS is NOT zero-initialised on hardware: poison it (0xFF.. = NaN doubles, -1 ints) so a
  // read of never-written LDS shows up as a wrong result here
  std::vector<char> dyn(shmem + 64, char(0xFF));
  b.dyn_smem = reinterpret_cast<char*>((reinterpret_cast<uintptr_t>(dyn.data()) + 63) & ~uintptr_t(63));
  if (!t_stack_pool)
    t_stack_pool = new std::vector<char*>();
  while (t_stack_pool->size() < block.x)
    t_stack_pool->push_back(static_cast<char*>(std::malloc(STACK_BYTES)));
  std::vector<Fiber> fibers(block.x);
  b.fibers.resize(block.x);
  for (unsigned t = 0; t < block.x; ++t) {
    prepare_fiber(&fibers[t], t, (*t_stack_pool)[t]);
    b.fibers[t] = &fibers[t];
  }
  g_block = &b;
  g_blockIdx.x = bx;
  g_blockIdx.y = 0;
  g_blockIdx.z = 0;
  g_blockDim = block;
  g_gridDim = grid;
  t_body = &body;
  t_idle_spins = 0;
  b.cur = reverse_order() ? block.x - 1 : 0;
  hipemu_switch(&t_main_sp, fibers[b.cur].sp);
  if (b.alive != 0)
    fatal("block finished with live fibers");
  g_block = nullptr;
}
} // namespace

void
fatal(const char* msg)
{
  std::fprintf(stderr, "[hipemu] FATAL: %s\n", msg);
  std::fflush(stderr);
  std::abort();
}

void
yield_to_next()
{
  Block* b = g_block;
  unsigned me = b->cur;
  if (++t_idle_spins > 64ul * b->nthreads + 1024)
    fatal("deadlock: a barrier / wave collective was not reached by every live thread "
          "(divergent collective?)");
  int nx = next_alive(b, me);
  if (nx < 0 || unsigned(nx) == me)
    return;
  switch_to(b, me, nx);
}

void
block_barrier()
{
  Block* b = g_block;
  unsigned long gen = b->bar_gen;
  b->bar_count++;
  if (b->bar_count >= b->alive) {
    b->bar_count = 0;
    b->bar_gen++;
    progress();
  } else {
    while (b->bar_gen == gen)
      yield_to_next();
  }
}

void
wave_barrier()
{
  Block* b = g_block;
  Block::Wave& w = b->waves[b->cur / 64];
  unsigned long gen = w.gen;
  w.count++;
  if (w.count >= w.alive) {
    w.count = 0;
    w.gen++;
    progress();
  } else {
    while (w.gen == gen)
      yield_to_next();
  }
}

std::uint64_t
wave_exchange(std::uint64_t v, int src_lane, bool)
{
  Block* b = g_block;
  unsigned me = b->cur;
  Block::Wave& w = b->waves[me / 64];
  unsigned lane = me % 64;
  w.slot64[lane] = v;
  wave_barrier();
  unsigned src_tid = (me / 64) * 64 + unsigned(src_lane & 63);
  std::uint64_t out = v;
  if (src_tid < b->nthreads && !b->fibers[src_tid]->done)
    out = w.slot64[unsigned(src_lane) & 63];
  wave_barrier();
  return out;
}

std::uint64_t
wave_ballot(bool pred)
{
  Block* b = g_block;
  unsigned me = b->cur;
  Block::Wave& w = b->waves[me / 64];
  unsigned lane = me % 64;
  w.slot64[lane] = pred ? 1 : 0;
  wave_barrier();
  std::uint64_t m = 0;
  unsigned base = (me / 64) * 64;
  for (unsigned l = 0; l < 64; ++l) {
    unsigned t = base + l;
    if (t < b->nthreads && !b->fibers[t]->done && w.slot64[l])
      m |= (std::uint64_t(1) << l);
  }
  wave_barrier();
  return m;
}

void
launch(dim3 grid, dim3 block, size_t shmem, const std::function<void()>& body)
{
  unsigned nblocks = grid.x;
  unsigned nthreads = 1;
  if (const char* e = std::getenv("HIPEMU_THREADS"))
    nthreads = unsigned(std::max(1, std::atoi(e)));
  else
    nthreads = std::max(1u, std::thread::hardware_concurrency());
  nthreads = std::min(nthreads, nblocks);
  if (nthreads <= 1) {
    for (unsigned bx = 0; bx < nblocks; ++bx)
      run_block(bx, grid, block, shmem, body);
    return;
  }
  std::atomic<unsigned> next{ 0 };
  std::vector<std::thread> pool;
  for (unsigned t = 0; t < nthreads; ++t)
    pool.emplace_back([&]() {
      while (true) {
        unsigned bx = next.fetch_add(1);
        if (bx >= nblocks)
          break;
        run_block(bx, grid, block, shmem, body);
      }
    });
  for (auto& th : pool)
    th.join();
}

} // namespace hipemu

double
hipemu_now_ms()
{
  using namespace std::chrono;
  return duration<double, std::milli>(steady_clock::now().time_since_epoch()).count();
}
