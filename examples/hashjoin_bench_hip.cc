// Host-resident drop-in benchmark: the reference's BM_HashMergeJoin loop (hashjoin_bench.cc:109-143)
// against include/hashmergejoin_hip.hpp, without google-benchmark.  Relations live in
// std::vector<std::pair<uint64_t,uint64_t>> in pageable host memory, exactly what a caller of the
// reference holds; every iteration constructs the join (PCIe in, GPU join, PCIe out) and reduces
// it.  Prints wall ms per join and the device-side phase split.  Host-only C++11 (g++).
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <utility>
#include <vector>

#include "hashmergejoin_hip.hpp"

typedef std::vector<std::pair<uint64_t, uint64_t>> KeyValVec;

static inline uint64_t mix64(uint64_t x) {
  x ^= x >> 30; x *= 0xBF58476D1CE4E5B9ull; x ^= x >> 27; x *= 0x94D049BB133111EBull; x ^= x >> 31;
  return x;
}

int main(int argc, char** argv) {
  int lo = argc > 1 ? atoi(argv[1]) : 20, hi = argc > 2 ? atoi(argv[2]) : 24;
  for (int lg = lo; lg <= hi; lg += 2) {
    const uint64_t n = 1ull << lg, seed = 0x243F6A8885A308D3ull;
    KeyValVec r(n), s(n);
    for (uint64_t i = 0; i < n; i++) {  // same key set, different order: every probe row matches once
      r[i] = std::make_pair(mix64(i + seed), i);
      uint64_t j = (0x9E3779B1ull * i + 12345) % n;
      s[i] = std::make_pair(mix64(j + seed), i ^ 0x9E3779B97F4A7C15ull);
    }
    hmj_ctx* c = hmj_detail::thread_ctx();
    hmj_set_profiling(c, 1);
    HashMergeJoin<KeyValVec::iterator, KeyValVec::iterator> hmj;
    double best = 1e30, best_iter = 0;
    hmj_timing tm{};
    uint64_t sum = 0;
    for (int it = 0; it < 4; it++) {
      hmj.clear();
      auto t0 = std::chrono::steady_clock::now();
      hmj = HashMergeJoin<KeyValVec::iterator, KeyValVec::iterator>(r.begin(), r.end(), s.begin(), s.end(),
                                                                     std::thread::hardware_concurrency());
      auto t1 = std::chrono::steady_clock::now();
      sum = 0;
      for (auto tuple : hmj) sum += *std::get<1>(tuple) + *std::get<2>(tuple);
      auto t2 = std::chrono::steady_clock::now();
      double ms = std::chrono::duration<double, std::milli>(t2 - t0).count();
      if (ms < best) {
        best = ms;
        best_iter = std::chrono::duration<double, std::milli>(t2 - t1).count();
        hmj_last_timing(c, &tm);
      }
    }
    std::printf("n=2^%d  ctor+iterate %.2f ms (iterate %.2f)  %.1f Mtuples/s | device: total %.2f h2d %.2f partition %.2f "
                "probe %.2f order %.2f d2h %.2f | rows %zu sum %llu\n",
                lg, best, best_iter, n / best / 1e3, tm.ms_total, tm.ms_h2d, tm.ms_hist + tm.ms_scatter + tm.ms_scan,
                tm.ms_probe_count + tm.ms_probe_write + tm.ms_out_scan, tm.ms_order, tm.ms_d2h, hmj.size(),
                (unsigned long long)sum);
  }
  return 0;
}
