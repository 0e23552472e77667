// GPU test + example of the C++ drop-in (include/hashmergejoin_hip.hpp): it is written the way
// the reference's own consumer is (hashjoin_bench.cc:109-143, BM_HashMergeJoin): build r and s as
// std::vector<std::pair<Key,uint64_t>>, construct HashMergeJoin from the iterator ranges,
// iterate, reduce.  Expected rows come from the CPU oracle (test infrastructure, linked only
// into this test binary).  Host-only C++11: built with g++, no hipcc.
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <deque>
#include <string>
#include <thread>
#include <utility>
#include <unordered_map>
#include <vector>

#include "hashmergejoin_hip.hpp"
#include "hmj_oracle.h"
#include "strgen_restated.h"

typedef std::vector<std::pair<uint64_t, uint64_t>> KeyValVec;  // reference: hashjoin.h:29 with Key=u64

static KeyValVec from_aos(const std::vector<uint64_t>& a) {
  KeyValVec v(a.size() / 2);
  for (size_t i = 0; i < v.size(); i++) v[i] = std::make_pair(a[2 * i], a[2 * i + 1]);
  return v;
}

static int run_case(uint64_t nb, uint64_t np, uint64_t miss, bool time_it) {
  std::vector<uint64_t> ba(2 * nb), pa(2 * np);
  orc_gen_build(ba.data(), nb, 0, ORC_SEED_B);
  orc_gen_probe(pa.data(), np, 0, nb ? nb : 1, ORC_SEED_B, miss);
  KeyValVec r = from_aos(ba), s = from_aos(pa);

  // reference semantics restated (pinned against the compiled reference in tests/)
  std::vector<uint64_t> want(3 * (np + 1));
  uint64_t want_sum = 0;
  uint64_t want_n = orc_hashmergejoin(ba.data(), nb, pa.data(), np, 2, want.data(), np + 1, &want_sum);

  HashMergeJoin<KeyValVec::iterator, KeyValVec::iterator> hmj;  // default ctor, hashjoin.h:55
  auto t0 = std::chrono::steady_clock::now();
  hmj = HashMergeJoin<KeyValVec::iterator, KeyValVec::iterator>(r.begin(), r.end(), s.begin(), s.end(),
                                                                 std::thread::hardware_concurrency());
  uint64_t sum = 0, n = 0;
  int bad = 0;
  for (auto tuple : hmj) {  // hashjoin_bench.cc:131-133
    sum += *std::get<1>(tuple) + *std::get<2>(tuple);
    if (n < want_n && (*std::get<0>(tuple) != want[3 * n] || *std::get<1>(tuple) != want[3 * n + 1] ||
                       *std::get<2>(tuple) != want[3 * n + 2]))
      bad++;
    n++;
  }
  double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  if (n != want_n || sum != want_sum || bad) {
    std::printf("FAIL nb=%llu np=%llu miss=%llu: n=%llu want %llu, sum=%llu want %llu, %d rows differ\n",
                (unsigned long long)nb, (unsigned long long)np, (unsigned long long)miss, (unsigned long long)n,
                (unsigned long long)want_n, (unsigned long long)sum, (unsigned long long)want_sum, bad);
    return 1;
  }
  // the free-function spelling and the sum-only reduction
  uint64_t n2 = 0;
  uint64_t sum2 = hash_merge_join_sum(r.begin(), r.end(), s.begin(), s.end(), &n2);
  auto j = join(r.begin(), r.end(), s.begin(), s.end());
  if (sum2 != want_sum || n2 != want_n || j.size() != want_n) {
    std::printf("FAIL sum-only/join(): %llu %llu\n", (unsigned long long)sum2, (unsigned long long)n2);
    return 1;
  }
  hmj.clear();  // hashjoin.h:192-195
  if (hmj.begin() != hmj.end()) return 1;
  if (time_it) {
    // again, now that the executor's pinned staging buffers and device arenas exist
    auto t1 = std::chrono::steady_clock::now();
    hmj = HashMergeJoin<KeyValVec::iterator, KeyValVec::iterator>(r.begin(), r.end(), s.begin(), s.end(),
                                                                   std::thread::hardware_concurrency());
    uint64_t sum3 = 0;
    for (auto tuple : hmj) sum3 += *std::get<1>(tuple) + *std::get<2>(tuple);
    double ms2 = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t1).count();
    if (sum3 != want_sum) return 1;
    std::printf("ok nb=%llu np=%llu: %llu rows, host-resident ctor+iterate %.2f ms first call, %.2f ms warm "
                "(PCIe included)\n", (unsigned long long)nb, (unsigned long long)np, (unsigned long long)n, ms, ms2);
  }
  return 0;
}

// ---- iterators other than std::vector<...>::iterator.  The reference only ever forms `begin + k`
// (radix_hash.h:375-388), so any random access iterator is a legal argument: a std::deque of pairs (NOT
// contiguous: its rows live in separate blocks), a reverse_iterator, a vector's const_iterator, a raw pointer.
// The header may hand the C ABI the caller's storage only for the last two kinds; the others must be staged.
template <typename RIt, typename SIt>
static int check_iter_pair(const char* what, RIt rb, RIt re, SIt sb, SIt se, const std::vector<uint64_t>& want,
                           uint64_t want_n, uint64_t want_sum) {
  HashMergeJoin<RIt, SIt> hmj(rb, re, sb, se, 2);
  uint64_t n = 0, sum = 0;
  int bad = 0;
  for (auto tuple : hmj) {
    sum += *std::get<1>(tuple) + *std::get<2>(tuple);
    if (n < want_n && (*std::get<0>(tuple) != want[3 * n] || *std::get<1>(tuple) != want[3 * n + 1] ||
                       *std::get<2>(tuple) != want[3 * n + 2]))
      bad++;
    n++;
  }
  uint64_t n2 = 0;
  const uint64_t sum2 = hash_merge_join_sum(rb, re, sb, se, &n2);
  if (n != want_n || sum != want_sum || bad || n2 != want_n || sum2 != want_sum) {
    std::printf("FAIL iterators (%s): n=%llu/%llu want %llu, sum=%llu/%llu want %llu, %d rows differ\n", what,
                (unsigned long long)n, (unsigned long long)n2, (unsigned long long)want_n, (unsigned long long)sum,
                (unsigned long long)sum2, (unsigned long long)want_sum, bad);
    return 1;
  }
  return 0;
}
static int run_iterator_kinds_case(uint64_t nb, uint64_t np, uint64_t miss) {
  std::vector<uint64_t> ba(2 * nb), pa(2 * np);
  orc_gen_build(ba.data(), nb, 0, ORC_SEED_B);
  orc_gen_probe(pa.data(), np, 0, nb ? nb : 1, ORC_SEED_B, miss);
  const KeyValVec r = from_aos(ba), s = from_aos(pa);
  std::vector<uint64_t> want(3 * (np + 1));
  uint64_t want_sum = 0;
  const uint64_t want_n = orc_hashmergejoin(ba.data(), nb, pa.data(), np, 2, want.data(), np + 1, &want_sum);
  typedef std::deque<std::pair<uint64_t, uint64_t>> KeyValDeque;
  static_assert(hmj_detail::is_hmj_relation_iter<KeyValDeque::iterator>::value &&
                    !hmj_detail::is_contiguous_iter<KeyValDeque::iterator>::value &&
                    !hmj_detail::is_contiguous_iter<KeyValVec::const_reverse_iterator>::value &&
                    hmj_detail::is_contiguous_iter<KeyValVec::const_iterator>::value &&
                    hmj_detail::is_contiguous_iter<KeyValVec::iterator>::value &&
                    hmj_detail::is_contiguous_iter<const std::pair<uint64_t, uint64_t>*>::value,
                "only pointers and std::vector iterators may be handed over as one address");
  KeyValDeque rd, sd;  // pushed at both ends so the rows straddle several of the deque's blocks
  for (uint64_t i = nb / 2; i < nb; i++) rd.push_back(r[i]);
  for (uint64_t i = nb / 2; i-- > 0;) rd.push_front(r[i]);
  for (uint64_t i = 0; i < np; i++) sd.push_back(s[i]);
  int fails = 0;
  fails += check_iter_pair("deque x deque", rd.begin(), rd.end(), sd.begin(), sd.end(), want, want_n, want_sum);
  fails += check_iter_pair("deque x vector", rd.begin(), rd.end(), s.begin(), s.end(), want, want_n, want_sum);
  fails += check_iter_pair("const_iterator", r.cbegin(), r.cend(), s.cbegin(), s.cend(), want, want_n, want_sum);
  fails += check_iter_pair("pointers", r.data(), r.data() + nb, s.data(), s.data() + np, want, want_n, want_sum);
  // a reversed view of a reversed copy is the same relation in the same order
  const KeyValVec rrev(r.rbegin(), r.rend()), srev(s.rbegin(), s.rend());
  fails += check_iter_pair("reverse_iterator", rrev.rbegin(), rrev.rend(), srev.rbegin(), srev.rend(), want, want_n,
                           want_sum);
  return fails;
}

// ---- a dimension table under a fact table: few build rows, hundreds of probe rows per key.  The operator's rows must come in
// (key, rval, sval) order -- the relational join's, which the oracle's orc_equijoin lists (the reference's own iterator is
// only defined for unique probe keys).  On the GPU this shape takes the sort on (key rank, payload) composites.
static int run_fk_case(uint64_t nb, uint64_t np) {
  std::vector<uint64_t> ba(2 * nb), pa(2 * np);
  orc_gen_build(ba.data(), nb, 0, ORC_SEED_B);
  orc_gen_uniform_domain(pa.data(), np, 0, nb, ORC_SEED_B, 0x7654321ull);
  const KeyValVec r = from_aos(ba), s = from_aos(pa);
  orc_checks ck;
  const uint64_t want_n = orc_equijoin(ba.data(), nb, pa.data(), np, 0, nullptr, 0, &ck);
  std::vector<uint64_t> want(3 * (want_n + 1));
  orc_equijoin(ba.data(), nb, pa.data(), np, 0, want.data(), want_n, &ck);
  HashMergeJoin<KeyValVec::const_iterator, KeyValVec::const_iterator> hmj(r.cbegin(), r.cend(), s.cbegin(), s.cend(), 2);
  uint64_t n = 0;
  int bad = 0;
  for (auto tuple : hmj) {
    if (n < want_n && (*std::get<0>(tuple) != want[3 * n] || *std::get<1>(tuple) != want[3 * n + 1] ||
                       *std::get<2>(tuple) != want[3 * n + 2]))
      bad++;
    n++;
  }
  if (n != want_n || bad) {
    std::printf("FAIL foreign-key case nb=%llu np=%llu: n=%llu want %llu, %d rows differ\n", (unsigned long long)nb,
                (unsigned long long)np, (unsigned long long)n, (unsigned long long)want_n, bad);
    return 1;
  }
  return 0;
}

// ---- std::string keys: the reference's own benchmark type (KeyValVec, hashjoin.h:29).  The same
// synthetic strgen-shaped relations as oracle/ref_driver.cc ref_hashmergejoin_str; the expected
// count / sum / ordered FNV come from the compiled reference (tests/golden/golden.json) and are
// compared by tests/test_gpu_join.py::test_cpp_dropin_operator from the STR lines printed here.
typedef std::vector<std::pair<std::string, uint64_t>> StrKeyValVec;
static std::string synth_key(uint64_t i, uint64_t seed) {
  return "w" + std::to_string(orc_mix64(i + seed) % 1000003ull) + "-" + std::to_string(i);
}
static void run_string_case(uint64_t nr, uint64_t ns, uint64_t seed) {
  StrKeyValVec r(nr), s(ns);
  for (uint64_t k = 0; k < nr; k++) {
    uint64_t i = (2654435761ull * k + 1) % nr;
    r[k] = std::make_pair(synth_key(i, seed), i);
  }
  for (uint64_t k = 0; k < ns; k++) {
    uint64_t i = (40503ull * k + 5) % ns;
    s[k] = std::make_pair(synth_key(nr / 2 + i, seed), 7 * i + 3);
  }
  HashMergeJoin<StrKeyValVec::iterator, StrKeyValVec::iterator> hmj(r.begin(), r.end(), s.begin(), s.end(), 4);
  uint64_t cnt = 0, sum = 0, fnv = 0xCBF29CE484222325ull;
  bool keys_ok = true;
  for (auto tuple : hmj) {
    const uint64_t rv = *std::get<1>(tuple), sv = *std::get<2>(tuple);
    sum += rv + sv;
    cnt++;
    keys_ok = keys_ok && *std::get<0>(tuple) == synth_key(rv, seed);  // the key handed out is the row's key
    const uint64_t w[2] = {rv, sv};
    for (int q = 0; q < 2; q++)
      for (int b = 0; b < 8; b++) {
        fnv ^= (w[q] >> (8 * b)) & 0xFF;
        fnv *= 0x100000001B3ull;
      }
  }
  std::printf("STR %llu %llu %llu count=%llu sum=%llu fnv=%llu keys_ok=%d\n", (unsigned long long)nr,
              (unsigned long long)ns, (unsigned long long)seed, (unsigned long long)cnt, (unsigned long long)sum,
              (unsigned long long)fnv, keys_ok ? 1 : 0);
}

// ---- std::string keys that REPEAT on the probe side (outside the reference's domain: its merge drops matches then,
// hashjoin.h:283-294): several result rows share a hash value, so the operator cannot take the GPU's columns as they are and
// walks the pairs one by one (hmj_detail::join_hashed_rows_visit's general walk, payloads gathered afterwards).  Expected:
// every equal-key pair once, ascending hash; count and sum by brute force, the key handed out is the row's key.
static int run_string_repeats_case(uint64_t nr, uint64_t ns, uint64_t seed) {
  StrKeyValVec r(nr), s(ns);
  for (uint64_t k = 0; k < nr; k++) r[k] = std::make_pair(synth_key(k, seed), k);
  for (uint64_t k = 0; k < ns; k++) s[k] = std::make_pair(synth_key((k * 7 + 1) % (nr + nr / 3 + 1), seed), 100 + k);  // wraps: keys repeat; some miss
  uint64_t want_cnt = 0, want_sum = 0;
  {
    std::unordered_map<std::string, uint64_t> m;
    for (uint64_t k = 0; k < nr; k++) m[r[k].first] = r[k].second;
    for (uint64_t k = 0; k < ns; k++) {
      auto it = m.find(s[k].first);
      if (it != m.end()) {
        want_cnt++;
        want_sum += it->second + s[k].second;
      }
    }
  }
  HashMergeJoin<StrKeyValVec::iterator, StrKeyValVec::iterator> hmj(r.begin(), r.end(), s.begin(), s.end(), 8);
  uint64_t cnt = 0, sum = 0, prev_hash = 0;
  bool ok = true;
  for (auto tuple : hmj) {
    const uint64_t rv = *std::get<1>(tuple), sv = *std::get<2>(tuple);
    const std::string& key = *std::get<0>(tuple);
    ok = ok && key == r[rv].first && key == s[sv - 100].first;
    const uint64_t h = std::hash<std::string>()(key);
    ok = ok && (cnt == 0 || h >= prev_hash);
    prev_hash = h;
    sum += rv + sv;
    cnt++;
  }
  ok = ok && cnt == want_cnt && sum == want_sum && hmj.size() == want_cnt;
  if (!ok)
    std::printf("string keys with repeats nr=%llu ns=%llu: count %llu (want %llu) sum %llu (want %llu) FAILED\n", (unsigned long long)nr,
                (unsigned long long)ns, (unsigned long long)cnt, (unsigned long long)want_cnt, (unsigned long long)sum,
                (unsigned long long)want_sum);
  return ok ? 0 : 1;
}

// ---- the reference's benchmark relations themselves: r = create_strvec(n), s = create_strvec(n)
// (hashjoin_bench.cc:112-113; strgen.cc:27-61 restated in oracle/strgen_restated.h over the word-list fixture
// tests/golden/words.txt).  Expected count / sum / ordered FNV: the compiled reference's (golden "strgen_join");
// n = 10^6 is BASELINE.json configs[0].
static void run_strgen_case(const std::vector<std::string>& words, int n) {
  StrKeyValVec r = hmj_strgen::create_strvec(n, words, 1), s = hmj_strgen::create_strvec(n, words, 2);
  const uint64_t fr = hmj_strgen::fnv_relation(r), fs = hmj_strgen::fnv_relation(s);
  auto t0 = std::chrono::steady_clock::now();
  HashMergeJoin<StrKeyValVec::iterator, StrKeyValVec::iterator> hmj(r.begin(), r.end(), s.begin(), s.end(),
                                                                     std::thread::hardware_concurrency());
  uint64_t cnt = 0, sum = 0, fnv = 0xCBF29CE484222325ull;
  bool keys_ok = true;
  const std::string* prev = nullptr;
  for (auto tuple : hmj) {
    const uint64_t rv = *std::get<1>(tuple), sv = *std::get<2>(tuple);
    sum += rv + sv;
    cnt++;
    // the key handed out belongs to a row of r with that payload pairing: rows of r and s with this key exist
    keys_ok = keys_ok && !std::get<0>(tuple)->empty() && std::get<0>(tuple) != prev;
    prev = std::get<0>(tuple);
    const uint64_t w[2] = {rv, sv};
    for (int q = 0; q < 2; q++)
      for (int b = 0; b < 8; b++) {
        fnv ^= (w[q] >> (8 * b)) & 0xFF;
        fnv *= 0x100000001B3ull;
      }
  }
  const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  std::printf("STRGEN %d count=%llu sum=%llu fnv=%llu keys_ok=%d fnv_r=%llu fnv_s=%llu ms=%.1f\n", n,
              (unsigned long long)cnt, (unsigned long long)sum, (unsigned long long)fnv, keys_ok ? 1 : 0,
              (unsigned long long)fr, (unsigned long long)fs, ms);
}

// ---- HashMergeJoin2 (hashjoin.h:201-363): pre-hashed std::tuple<hash, key, value> rows.
// mode 0: hash = key; mode 1: hash = mix64(key), a bijection -- both inside the reference's domain (one
// key per hash value): the rows must equal the restated reference's, in its order.
// mode 2: hash = key mod 997, many different keys per hash value.  There the reference's merge treats
// equal hashes like duplicate keys and DROPS matches (its staircase advance, hashjoin.h:283-294; the
// oracle reproduces that, tests/test_oracle_vs_ref.py); this operator returns the relational join --
// every pair of equal keys, ascending (hash, key) -- and that is what is checked.
typedef std::vector<std::tuple<std::size_t, uint64_t, uint64_t>> HashKeyValVec;  // hashjoin.h:30-31
static int run_prehashed_case(uint64_t nb, uint64_t np, uint64_t miss, int mode) {
  std::vector<uint64_t> ba(2 * nb), pa(2 * np);
  orc_gen_build(ba.data(), nb, 0, ORC_SEED_B);
  orc_gen_probe(pa.data(), np, 0, nb ? nb : 1, ORC_SEED_B, miss);
  auto hash_of = [mode](uint64_t k) { return mode == 0 ? k : mode == 1 ? orc_mix64(k) : k % 997; };
  HashKeyValVec r(nb), s(np);
  std::vector<uint64_t> rh(3 * nb), sh(3 * np);
  for (uint64_t i = 0; i < nb; i++) {
    const uint64_t h = hash_of(ba[2 * i]);
    r[i] = std::make_tuple((std::size_t)h, ba[2 * i], ba[2 * i + 1]);
    rh[3 * i] = h; rh[3 * i + 1] = ba[2 * i]; rh[3 * i + 2] = ba[2 * i + 1];
  }
  for (uint64_t i = 0; i < np; i++) {
    const uint64_t h = hash_of(pa[2 * i]);
    s[i] = std::make_tuple((std::size_t)h, pa[2 * i], pa[2 * i + 1]);
    sh[3 * i] = h; sh[3 * i + 1] = pa[2 * i]; sh[3 * i + 2] = pa[2 * i + 1];
  }
  std::vector<uint64_t> want(3 * (np + 1));
  uint64_t want_sum = 0, want_n;
  if (mode < 2) {
    want_n = orc_hashmergejoin2(rh.data(), nb, sh.data(), np, want.data(), np + 1, &want_sum);
  } else {
    orc_checks ck;
    want_n = orc_equijoin(ba.data(), nb, pa.data(), np, 0, nullptr, 0, &ck);
    want_sum = ck.sum_r + ck.sum_s;
  }
  HashMergeJoin2<HashKeyValVec::iterator, HashKeyValVec::iterator> hmj;
  hmj = HashMergeJoin2<HashKeyValVec::iterator, HashKeyValVec::iterator>(r.begin(), r.end(), s.begin(), s.end(), 4);
  uint64_t n = 0, sum = 0, prev_h = 0, prev_k = 0;
  int bad = 0;
  for (auto tuple : hmj) {
    const uint64_t k = *std::get<0>(tuple);
    sum += *std::get<1>(tuple) + *std::get<2>(tuple);
    if (mode < 2) {
      if (n < want_n && (k != want[3 * n] || *std::get<1>(tuple) != want[3 * n + 1] ||
                         *std::get<2>(tuple) != want[3 * n + 2]))
        bad++;
    } else {  // ascending (hash, key); the payloads are the rows' own
      const uint64_t h = hash_of(k);
      if (n && (h < prev_h || (h == prev_h && k <= prev_k))) bad++;
      if (*std::get<1>(tuple) >= nb || ba[2 * *std::get<1>(tuple)] != k) bad++;
      prev_h = h;
      prev_k = k;
    }
    n++;
  }
  if (n != want_n || sum != want_sum || bad) {
    std::printf("FAIL HashMergeJoin2 nb=%llu np=%llu mode=%d: n=%llu want %llu, %d rows differ\n", (unsigned long long)nb,
                (unsigned long long)np, mode, (unsigned long long)n, (unsigned long long)want_n, bad);
    return 1;
  }
  // The ctor's side effect (hashjoin.h:234-235 -> radix_inplace_par): the CALLER'S buffers are now sorted by
  // hash.  Inside the reference's domain (one key per hash value; modes 0 and 1) the sorted buffers are unique:
  // they must equal what the restated reference left in rh / sh.  Mode 2 (many keys per hash): ascending
  // hashes over the same multiset of rows, equal hashes in input order.
  {
    int unsorted = 0;
    const HashKeyValVec* bufs[2] = {&r, &s};
    const std::vector<uint64_t>* wants[2] = {&rh, &sh};
    for (int q = 0; q < 2; q++) {
      const HashKeyValVec& v = *bufs[q];
      uint64_t xr = 0, xw = 0;
      for (size_t i = 0; i < v.size(); i++) {
        const uint64_t h = std::get<0>(v[i]), k = std::get<1>(v[i]), p = std::get<2>(v[i]);
        if (i && h < (uint64_t)std::get<0>(v[i - 1])) unsorted++;
        if (mode < 2 && (h != (*wants[q])[3 * i] || k != (*wants[q])[3 * i + 1] || p != (*wants[q])[3 * i + 2])) unsorted++;
        if (mode == 2 && i && h == (uint64_t)std::get<0>(v[i - 1]) && q == 0 && p < std::get<2>(v[i - 1])) unsorted++;  // build payload = input position
        xr ^= orc_mix64(h ^ orc_mix64(k ^ orc_mix64(p)));
        xw ^= orc_mix64((*wants[q])[3 * i] ^ orc_mix64((*wants[q])[3 * i + 1] ^ orc_mix64((*wants[q])[3 * i + 2])));
      }
      if (xr != xw) unsorted++;  // same multiset of rows
    }
    if (unsorted) {
      std::printf("FAIL HashMergeJoin2 nb=%llu np=%llu mode=%d: caller buffers not sorted in place (%d)\n",
                  (unsigned long long)nb, (unsigned long long)np, mode, unsorted);
      return 1;
    }
  }
  hmj.clear();
  return hmj.begin() != hmj.end();
}

// HashMergeJoin2 over tuples the GPU cannot move (std::string keys): the buffers are permuted on the host into
// the order hmj_argsort_u64_host computes; same contract (sorted by hash afterwards, every key pairs once).
static int run_prehashed_string_case(const std::vector<std::string>& words, int n) {
  typedef std::vector<std::tuple<std::size_t, std::string, uint64_t>> HashStrVec;
  StrKeyValVec a = hmj_strgen::create_strvec(n, words, 5), b = hmj_strgen::create_strvec(n, words, 6);
  HashStrVec r(n), s(n);
  std::hash<std::string> h;
  uint64_t want_sum = 0;
  for (int i = 0; i < n; i++) {
    r[i] = std::make_tuple(h(a[i].first), a[i].first, a[i].second);
    s[i] = std::make_tuple(h(b[i].first), b[i].first, b[i].second);
    want_sum += a[i].second + b[i].second;
  }
  HashMergeJoin2<HashStrVec::iterator, HashStrVec::iterator> hmj(r.begin(), r.end(), s.begin(), s.end(), 2);
  uint64_t cnt = 0, sum = 0;
  for (auto t : hmj) {
    sum += *std::get<1>(t) + *std::get<2>(t);
    cnt++;
  }
  int bad = 0;
  for (int i = 1; i < n; i++)
    if (std::get<0>(r[i]) < std::get<0>(r[i - 1]) || std::get<0>(s[i]) < std::get<0>(s[i - 1])) bad++;
  for (int i = 0; i < n; i++)
    if (std::get<0>(r[i]) != h(std::get<1>(r[i])) || std::get<0>(s[i]) != h(std::get<1>(s[i]))) bad++;  // rows stayed whole
  if (cnt != (uint64_t)n || sum != want_sum || bad) {
    std::printf("FAIL HashMergeJoin2<string> n=%d: count=%llu sum=%llu want %llu, %d order errors\n", n,
                (unsigned long long)cnt, (unsigned long long)sum, (unsigned long long)want_sum, bad);
    return 1;
  }
  return 0;
}

int main(int argc, char** argv) {
  int fails = 0;
  fails += run_case(0, 0, 0, false);
  fails += run_case(0, 7, 0, false);
  fails += run_case(9, 0, 0, false);
  fails += run_case(1, 1, 0, false);
  fails += run_case(1000, 1000, 0, false);
  fails += run_case(5000, 3000, 3, false);
  fails += run_case(1 << 16, 1 << 16, 2, false);
  fails += run_case(1 << 20, 1 << 20, 0, true);
  fails += run_case(1 << 22, 1 << 22, 0, true);
  fails += run_iterator_kinds_case(0, 0, 0);
  fails += run_iterator_kinds_case(1, 1, 0);
  fails += run_iterator_kinds_case(5000, 3000, 3);
  fails += run_iterator_kinds_case(300000, 200000, 2);
  fails += run_fk_case(1, 3000);
  fails += run_fk_case(1000, 300000);
  fails += run_fk_case(500, 1 << 22);  // (thousands of rows per key over a few million probe rows: the sort on composites)
  fails += run_fk_case(30000, 200000);  // (fan-out 6: the partitioned one-pass ordered write)
  fails += run_prehashed_case(0, 5, 0, 0);
  fails += run_prehashed_case(1000, 1000, 0, 0);
  fails += run_prehashed_case(5000, 3000, 3, 1);
  fails += run_prehashed_case(1 << 16, 1 << 16, 2, 1);
  fails += run_prehashed_case(300000, 200000, 0, 1);
  fails += run_prehashed_case(5000, 3000, 3, 2);
  fails += run_prehashed_case(1 << 16, 1 << 16, 2, 2);
  {  // seeded sweep over sizes around the executor's tile / table boundaries
    uint64_t x = 88172645463325252ull;
    auto next = [&x]() { x ^= x << 13; x ^= x >> 7; x ^= x << 17; return x; };
    const uint64_t sizes[] = {1, 2, 63, 64, 65, 2047, 2048, 2049, 4095, 4096, 4097, 5119, 5120, 5121, 8191, 20000, 65537, 150001};
    for (int it = 0; it < 40; it++) {
      const uint64_t nb = sizes[next() % 18], np = sizes[next() % 18];
      fails += run_case(nb, np > nb ? nb : np, next() % 4, false);  // unique probe keys need np <= nb
      if (it % 4 == 0) fails += run_prehashed_case(nb, np > nb ? nb : np, next() % 4, 1);
    }
  }
  run_string_case(1000, 1000, 1);
  run_string_case(5000, 3000, 2);
  run_string_case(200000, 150000, 3);
  fails += run_string_repeats_case(10, 100, 4);
  fails += run_string_repeats_case(3000, 20000, 5);
  fails += run_string_repeats_case(100000, 400000, 6);
  {
    const std::string words_path = argc > 1 ? argv[1] : "tests/golden/words.txt";
    const std::vector<std::string> words = hmj_strgen::load_words(words_path);
    const int sizes[] = {2, 1000, 1 << 12, 1 << 16, 1 << 18, 1000000};
    for (int n : sizes) run_strgen_case(words, n);
    fails += run_prehashed_string_case(words, 1000);
    fails += run_prehashed_string_case(words, 70000);
  }
  std::printf(fails ? "FAILED\n" : "all drop-in cases passed\n");
  return fails ? 1 : 0;
}
