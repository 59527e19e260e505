// stress_pool.cpp -- ThreadSanitizer stress of ckm::HostPool (checkm_amd/csrc/host_pool.h), built by tests/test_native_sanitize.py.
// Many short jobs back to back (the pattern of the cascade: a pool_run per stage, microseconds apart), jobs smaller than one
// chunk, a job that throws, two pools side by side (two workers), and destruction while helpers are idle.
#include <cstdio>
#include <numeric>
#include <stdexcept>
#include <vector>
#include "host_pool.h"

static int one_pool(int nthreads, int rounds) {
  ckm::HostPool pool(nthreads);
  std::vector<long> out;
  for (int r = 0; r < rounds; ++r) {
    const size_t n = 1 + (size_t)(r * 37 % 5000), chunk = 1 + (size_t)(r % 64);
    out.assign(n, 0);
    pool.run(n, chunk, [&](size_t lo, size_t hi) { for (size_t i = lo; i < hi; ++i) out[i] = (long)i * 3 + r; });
    for (size_t i = 0; i < n; ++i) if (out[i] != (long)i * 3 + r) { fprintf(stderr, "round %d: element %zu not written\n", r, i); return 1; }
    if (r % 97 == 0) {
      bool thrown = false;
      try { pool.run(1000, 10, [&](size_t lo, size_t hi) { if (lo <= 500 && 500 < hi) throw std::runtime_error("boom"); }); } catch (const std::runtime_error &) { thrown = true; }
      if (!thrown) { fprintf(stderr, "exception of a chunk was lost\n"); return 1; }
    }
  }
  return 0;
}

int main() {
  int rc = 0;
  std::thread a([&] { if (one_pool(8, 3000)) rc = 1; }), b([&] { if (one_pool(3, 3000)) rc = 1; });
  a.join(); b.join();
  if (one_pool(1, 50)) rc = 1;        // no helpers: the caller does everything
  printf("%s\n", rc ? "FAILED" : "ok");
  return rc;
}
