// Worker threads of the host twins (host_twins.hip, estimators_host.hip): plain std::thread over contiguous ranges,
// at most lsr_set_host_threads(n) of them (default 1) -- no OpenMP runtime enters the process (the reference's warning
// about torch plus a second OpenMP, shrimpy/tests/conftest.py:11-17).
#pragma once

#include <atomic>
#include <cstdint>
#include <new>
#include <system_error>
#include <thread>
#include <vector>

namespace lsr {

inline std::atomic<int> g_host_threads{1};
inline std::atomic<bool> g_host_dummy_failed{false};

// fn(k, first, last) over [0, n) split into contiguous ranges, one per worker; k = the range's rank (0, 1, ... in
// order of `first`), so a reduction that keeps one partial result per k and adds them up in k order does not depend
// on scheduling.  Nothing may escape a worker (an exception leaving a std::thread ends the process): a range that
// cannot get its scratch memory reports through `failed` and the entry point returns an error.  Returns the number
// of ranges (<= lsr_get_host_threads() <= 1024).
template <typename F>
int parallel_ranges_indexed(int64_t n, F&& fn_raw, std::atomic<bool>& failed = g_host_dummy_failed) {
  auto fn = [&fn_raw, &failed](int k, int64_t a, int64_t b) {
    try {
      fn_raw(k, a, b);
    } catch (const std::bad_alloc&) {
      failed.store(true, std::memory_order_relaxed);
    }
  };
  int workers = g_host_threads.load(std::memory_order_relaxed);
  if (workers > n) workers = static_cast<int>(n);
  if (workers <= 1) {
    fn(0, int64_t(0), n);
    return 1;
  }
  std::vector<std::thread> pool;
  pool.reserve(static_cast<size_t>(workers));
  const int64_t per = (n + workers - 1) / workers;
  int used = 0;
  for (int w = 0; w < workers; ++w) {
    const int64_t a = w * per, b = a + per < n ? a + per : n;
    if (a >= b) break;
    const int k = used++;
    try {
      pool.emplace_back([&fn, k, a, b] { fn(k, a, b); });
    } catch (const std::system_error&) {   // the box refuses another thread: this range runs here
      fn(k, a, b);
    }
  }
  for (std::thread& t : pool) t.join();
  return used;
}

template <typename F>
int parallel_ranges(int64_t n, F&& fn, std::atomic<bool>& failed = g_host_dummy_failed) {
  return parallel_ranges_indexed(n, [&fn](int, int64_t a, int64_t b) { fn(a, b); }, failed);
}

// The twins' translation units are built with the host's FMA3 instructions enabled (the stencils' explicit fmaf
// chains then cost one instruction each instead of a libm call); api.hip -- built without -- answers whether the
// CPU has them, and every twin asks before it runs anything.
bool host_fma_ok();

}  // namespace lsr

#define LSR_REQUIRE_HOST_FMA()                                                                                 \
  LSR_REQUIRE(lsr::host_fma_ok(), LSR_E_UNSUPPORTED, "this CPU has no FMA3 instructions: the host twins need them")
