// TEST INFRASTRUCTURE ONLY -- not part of the product path.
//
// C-ABI driver around the *real* reference headers.  This file is our own code; it
// #includes hashjoin.h / radix_hash.h / radix_sort.h / partitioned_hash.h from the
// read-only reference tree at build time (-I$(REF), see oracle/Makefile) and is
// compiled only into oracle/_ref/libhmj_ref.so (git-ignored).  No reference source
// is copied into this repository.
//
// Uses:  (1) validate the C restatement in hmj_oracle.c,
//        (2) generate tests/golden/* (tests/golden/make_golden.py),
//        (3) bench.py's cpu_baseline leg, kind "reference".
//
// Relation layout everywhere: n x {uint64 key; uint64 val} == std::pair<u64,u64>
// (SURVEY.md D4 / H6).  std::tuple is stored in reverse order by libstdc++, so
// tuples are always (un)packed with std::get, never memcpy'd.

#include <chrono>
#include <cstdint>
#include <cstring>
#include <tuple>
#include <unordered_map>
#include <unordered_set>
#include <utility>
#include <vector>

#include "strgen_restated.h"
#include "hashjoin.h"
#include "partitioned_hash.h"
#include "radix_hash.h"
#include "radix_sort.h"

typedef std::vector<std::pair<uint64_t, uint64_t>> PairVec;
typedef std::vector<std::tuple<std::size_t, uint64_t, uint64_t>> TupVec;

static PairVec to_pairs(const uint64_t* aos, uint64_t n) {
  PairVec v(n);
  for (uint64_t i = 0; i < n; i++) v[i] = std::make_pair(aos[2 * i], aos[2 * i + 1]);
  return v;
}

extern "C" {

// radix_hash.h:38-57
int ref_optimal_partition(uint64_t n) { return radix_hash::optimal_partition(n); }

// hashjoin.h:56-68 (ctor) + :104-173 (iteration), as driven by hashjoin_bench.cc:126-133.
// Writes at most `cap` triples {key,rval,sval}; returns the number of tuples yielded.
// *sum_out receives the bench's own reduction, sum += rval + sval.
uint64_t ref_hashmergejoin_u64(const uint64_t* r_aos, uint64_t nr, const uint64_t* s_aos,
                               uint64_t ns, unsigned threads, uint64_t* triples, uint64_t cap,
                               uint64_t* sum_out) {
  PairVec r = to_pairs(r_aos, nr), s = to_pairs(s_aos, ns);
  HashMergeJoin<PairVec::iterator, PairVec::iterator> hmj(r.begin(), r.end(), s.begin(),
                                                          s.end(), threads);
  uint64_t cnt = 0, sum = 0;
  for (auto t : hmj) {
    if (cnt < cap && triples) {
      triples[3 * cnt + 0] = *std::get<0>(t);
      triples[3 * cnt + 1] = *std::get<1>(t);
      triples[3 * cnt + 2] = *std::get<2>(t);
    }
    sum += *std::get<1>(t) + *std::get<2>(t);
    cnt++;
  }
  if (sum_out) *sum_out = sum;
  return cnt;
}

// Same, timing only the part hashjoin_bench.cc:125-134 times; inputs prebuilt by caller.
// (the ctypes caller times this call; conversion to PairVec is done by ref_pairs_* below)
void* ref_pairs_new(const uint64_t* aos, uint64_t n) { return new PairVec(to_pairs(aos, n)); }
void ref_pairs_free(void* p) { delete static_cast<PairVec*>(p); }
uint64_t ref_hashmergejoin_pairs(void* rp, void* sp, unsigned threads, uint64_t* sum_out) {
  PairVec& r = *static_cast<PairVec*>(rp);
  PairVec& s = *static_cast<PairVec*>(sp);
  HashMergeJoin<PairVec::iterator, PairVec::iterator> hmj(r.begin(), r.end(), s.begin(),
                                                          s.end(), threads);
  uint64_t cnt = 0, sum = 0;
  for (auto t : hmj) {
    sum += *std::get<1>(t) + *std::get<2>(t);
    cnt++;
  }
  if (sum_out) *sum_out = sum;
  return cnt;
}

// radix_hash.h:351-406 (explicit bits) / :408-423 (bits<0 -> optimal_partition).
// out: n x {hash,key,val}.
void ref_radix_non_inplace_par_u64(const uint64_t* aos, uint64_t n, int threads, int bits,
                                   uint64_t* out) {
  PairVec src = to_pairs(aos, n);
  TupVec dst(n);
  if (bits < 0)
    radix_hash::radix_non_inplace_par<uint64_t, uint64_t>(src.begin(), src.end(), dst.begin(),
                                                          threads);
  else
    radix_hash::radix_non_inplace_par<uint64_t, uint64_t>(src.begin(), src.end(), dst.begin(),
                                                          threads, bits);
  for (uint64_t i = 0; i < n; i++) {
    out[3 * i + 0] = std::get<0>(dst[i]);
    out[3 * i + 1] = std::get<1>(dst[i]);
    out[3 * i + 2] = std::get<2>(dst[i]);
  }
}

// radix_hash.h:425-493.  inout: n x {hash,key,val}.
void ref_radix_inplace_seq_u64(uint64_t* hkv, uint64_t n, int bits) {
  TupVec dst(n);
  for (uint64_t i = 0; i < n; i++)
    dst[i] = std::make_tuple((std::size_t)hkv[3 * i], hkv[3 * i + 1], hkv[3 * i + 2]);
  if (bits < 0)
    radix_hash::radix_inplace_seq<uint64_t, uint64_t>(dst.begin(), n);
  else
    radix_hash::radix_inplace_seq<uint64_t, uint64_t>(dst.begin(), n, bits);
  for (uint64_t i = 0; i < n; i++) {
    hkv[3 * i + 0] = std::get<0>(dst[i]);
    hkv[3 * i + 1] = std::get<1>(dst[i]);
    hkv[3 * i + 2] = std::get<2>(dst[i]);
  }
}

// radix_hash.h:589-654.  inout: n x {hash,key,val}.
void ref_radix_inplace_par_u64(uint64_t* hkv, uint64_t n, int threads, int bits) {
  TupVec dst(n);
  for (uint64_t i = 0; i < n; i++)
    dst[i] = std::make_tuple((std::size_t)hkv[3 * i], hkv[3 * i + 1], hkv[3 * i + 2]);
  if (bits < 0)
    radix_hash::radix_inplace_par(dst.begin(), n, threads);
  else
    radix_hash::radix_inplace_par(dst.begin(), n, threads, bits);
  for (uint64_t i = 0; i < n; i++) {
    hkv[3 * i + 0] = std::get<0>(dst[i]);
    hkv[3 * i + 1] = std::get<1>(dst[i]);
    hkv[3 * i + 2] = std::get<2>(dst[i]);
  }
}

// radix_sort.h:452-522 -- what radix_bench_par.cc:126-127 times.  out: n x {key,val}.
void ref_radix_int_non_inplace_u64(const uint64_t* aos, uint64_t n, int threads, int bits,
                                   uint64_t* out) {
  PairVec src = to_pairs(aos, n);
  PairVec dst(n);
  if (bits < 0)
    ::radix_int_non_inplace<uint64_t, uint64_t>(src.begin(), src.end(), dst.begin(), threads);
  else
    ::radix_int_non_inplace<uint64_t, uint64_t>(src.begin(), src.end(), dst.begin(), threads,
                                                bits);
  for (uint64_t i = 0; i < n; i++) {
    out[2 * i] = dst[i].first;
    out[2 * i + 1] = dst[i].second;
  }
}
void ref_radix_int_non_inplace_pairs(void* inp, void* outp, int threads) {
  PairVec& in = *static_cast<PairVec*>(inp);
  PairVec& out = *static_cast<PairVec*>(outp);
  ::radix_int_non_inplace<uint64_t, uint64_t>(in.begin(), in.end(), out.begin(), threads);
}

// radix_sort.h:333-398 -- what radix_bench_par.cc:96 times.  inout: n x {key,val}.
void ref_radix_int_inplace_u64(uint64_t* aos, uint64_t n, int threads, int bits) {
  PairVec w = to_pairs(aos, n);
  if (bits < 0)
    ::radix_int_inplace<uint64_t, uint64_t>(w.begin(), (unsigned int)n, threads);
  else
    ::radix_int_inplace<uint64_t, uint64_t>(w.begin(), (std::size_t)n, threads, bits);
  for (uint64_t i = 0; i < n; i++) {
    aos[2 * i] = w[i].first;
    aos[2 * i + 1] = w[i].second;
  }
}

// partitioned_hash.h:82-124.  sizes_out[2^bits]; if content_out != NULL it receives the
// n tuples bucket after bucket (deterministic only for threads == 1).
void ref_partition_only_u64(const uint64_t* aos, uint64_t n, int threads, int bits,
                            uint64_t* sizes_out, uint64_t* content_out) {
  PairVec src = to_pairs(aos, n);
  std::vector<PairVec> dst(1u << bits);
  radix_hash::partition_only(src.begin(), src.end(), &dst, threads, bits);
  uint64_t k = 0;
  for (size_t p = 0; p < dst.size(); p++) {
    sizes_out[p] = dst[p].size();
    if (content_out)
      for (auto& kv : dst[p]) {
        content_out[2 * k] = kv.first;
        content_out[2 * k + 1] = kv.second;
        k++;
      }
  }
}

// partitioned_hash.h:173-215.  sizes_out[2^bits] = tables[p].size().
void ref_partitioned_hash_table_sizes_u64(const uint64_t* aos, uint64_t n, int threads,
                                          int bits, uint64_t* sizes_out) {
  PairVec src = to_pairs(aos, n);
  std::vector<std::unordered_map<uint64_t, uint64_t>> tables(1u << bits);
  radix_hash::partitioned_hash_table(src.begin(), src.end(), &tables, threads, bits);
  for (size_t p = 0; p < tables.size(); p++) sizes_out[p] = tables[p].size();
}

// The partition + build + probe formulation exactly as hashjoin_bench.cc:88-96 drives it
// (that bench cannot be compiled here: google-benchmark absent):
//   partition_only(probe side), partitioned_hash_table(build side), then the serial loop
//   sum += probe.val + tables[p][probe.key]   (operator[]: miss inserts 0).
// n_found (optional) = probes whose key was present before the lookup.
uint64_t ref_partitioned_join_sum_u64(const uint64_t* probe_aos, uint64_t n_probe,
                                      const uint64_t* build_aos, uint64_t n_build, int threads,
                                      int bits, uint64_t* n_found) {
  PairVec r = to_pairs(probe_aos, n_probe), s = to_pairs(build_aos, n_build);
  std::vector<PairVec> r_vectors(1u << bits);
  std::vector<std::unordered_map<uint64_t, uint64_t>> s_tables(1u << bits);
  radix_hash::partition_only(r.begin(), r.end(), &r_vectors, threads, bits);
  radix_hash::partitioned_hash_table(s.begin(), s.end(), &s_tables, threads, bits);
  uint64_t sum = 0, found = 0;
  for (size_t i = 0; i < r_vectors.size(); ++i) {
    for (auto r_pair : r_vectors[i]) {
      found += s_tables[i].count(r_pair.first);
      sum += r_pair.second + s_tables[i][r_pair.first];
    }
  }
  if (n_found) *n_found = found;
  return sum;
}

// hashjoin.h:201-363: join over caller-provided pre-hashed (hash,key,val) buffers, sorted
// in place (radix_inplace_par).  r_hkv / s_hkv are overwritten with the sorted tuples.
uint64_t ref_hashmergejoin2_u64(uint64_t* r_hkv, uint64_t nr, uint64_t* s_hkv, uint64_t ns,
                                unsigned threads, uint64_t* triples, uint64_t cap,
                                uint64_t* sum_out) {
  TupVec r(nr), s(ns);
  for (uint64_t i = 0; i < nr; i++)
    r[i] = std::make_tuple((std::size_t)r_hkv[3 * i], r_hkv[3 * i + 1], r_hkv[3 * i + 2]);
  for (uint64_t i = 0; i < ns; i++)
    s[i] = std::make_tuple((std::size_t)s_hkv[3 * i], s_hkv[3 * i + 1], s_hkv[3 * i + 2]);
  HashMergeJoin2<TupVec::iterator, TupVec::iterator> hmj(r.begin(), r.end(), s.begin(), s.end(),
                                                         threads);
  uint64_t cnt = 0, sum = 0;
  for (auto t : hmj) {
    if (cnt < cap && triples) {
      triples[3 * cnt + 0] = *std::get<0>(t);
      triples[3 * cnt + 1] = *std::get<1>(t);
      triples[3 * cnt + 2] = *std::get<2>(t);
    }
    sum += *std::get<1>(t) + *std::get<2>(t);
    cnt++;
  }
  if (sum_out) *sum_out = sum;
  return cnt;
}

// String keys: the reference's own benchmark type (KeyValVec = vector<pair<string,uint64_t>>,
// hashjoin.h:29; hashjoin_bench.cc:109-143).  strgen needs /usr/share/dict/words (absent), so the
// relations come from a synthetic generator of the same shape ("word-word" keys, unique, shuffled);
// tests/cpp/test_dropin.cc generates the identical relations.  Writes at most cap (rval, sval) pairs
// in iteration order; returns the number of tuples yielded.
static inline uint64_t drv_mix64(uint64_t x) {
  x ^= x >> 30; x *= 0xBF58476D1CE4E5B9ull; x ^= x >> 27; x *= 0x94D049BB133111EBull; x ^= x >> 31;
  return x;
}
static std::string drv_synth_key(uint64_t i, uint64_t seed) {
  return "w" + std::to_string(drv_mix64(i + seed) % 1000003ull) + "-" + std::to_string(i);
}
uint64_t ref_hashmergejoin_str(uint64_t nr, uint64_t ns, uint64_t seed, unsigned threads, uint64_t* pairs,
                               uint64_t cap, uint64_t* sum_out) {
  KeyValVec r(nr), s(ns);
  for (uint64_t k = 0; k < nr; k++) {
    uint64_t i = (2654435761ull * k + 1) % nr;
    r[k] = std::make_pair(drv_synth_key(i, seed), i);
  }
  for (uint64_t k = 0; k < ns; k++) {
    uint64_t i = (40503ull * k + 5) % ns;
    s[k] = std::make_pair(drv_synth_key(nr / 2 + i, seed), 7 * i + 3);
  }
  HashMergeJoin<KeyValVec::iterator, KeyValVec::iterator> hmj(r.begin(), r.end(), s.begin(), s.end(), threads);
  uint64_t cnt = 0, sum = 0;
  for (auto t : hmj) {
    if (pairs && cnt < cap) {
      pairs[2 * cnt] = *std::get<1>(t);
      pairs[2 * cnt + 1] = *std::get<2>(t);
    }
    sum += *std::get<1>(t) + *std::get<2>(t);
    cnt++;
  }
  if (sum_out) *sum_out = sum;
  return cnt;
}

// The reference's benchmark relations (hashjoin_bench.cc:112-113: r = create_strvec(n), s = create_strvec(n)) from
// the restated generator (strgen_restated.h: word list as a parameter, seeded shuffle), joined by the REAL
// HashMergeJoin<KeyValVec::iterator, KeyValVec::iterator> (hashjoin.h:33-199).  fnv_rel_out[0..1]: FNV-1a of
// the two generated relations (pins the generator); pairs: (rval, sval) in iteration order.
uint64_t ref_hashmergejoin_strgen(const char* words_path, uint64_t n, uint64_t seed_r, uint64_t seed_s,
                                  unsigned threads, uint64_t* pairs, uint64_t cap, uint64_t* sum_out,
                                  uint64_t* fnv_rel_out, uint64_t* distinct_out) {
  const std::vector<std::string> words = hmj_strgen::load_words(words_path);
  KeyValVec r = hmj_strgen::create_strvec((int)n, words, seed_r), s = hmj_strgen::create_strvec((int)n, words, seed_s);
  if (fnv_rel_out) {
    fnv_rel_out[0] = hmj_strgen::fnv_relation(r);
    fnv_rel_out[1] = hmj_strgen::fnv_relation(s);
  }
  if (distinct_out) {  // strgen_test.cc:24-33: all keys distinct
    std::unordered_set<std::string> u;
    for (const auto& p : r) u.insert(p.first);
    *distinct_out = u.size();
  }
  HashMergeJoin<KeyValVec::iterator, KeyValVec::iterator> hmj(r.begin(), r.end(), s.begin(), s.end(), threads);
  uint64_t cnt = 0, sum = 0;
  for (auto t : hmj) {
    if (pairs && cnt < cap) {
      pairs[2 * cnt] = *std::get<1>(t);
      pairs[2 * cnt + 1] = *std::get<2>(t);
    }
    sum += *std::get<1>(t) + *std::get<2>(t);
    cnt++;
  }
  if (sum_out) *sum_out = sum;
  return cnt;
}

// The same join TIMED as hashjoin_bench.cc:120-134 times it (BM_HashMergeJoin): relations generated before the clock
// starts (the benchmark pauses its timer for create_strvec), then `reps` x (construct + iterate + reduce), best time
// returned in seconds; count / sum / ordered FNV of the (rval, sval) pairs of the last repetition in out3[0..2].
// bench.py's cpu_baseline leg: BASELINE.json configs[0] next to the GPU drop-in on the same relations.
double ref_hashmergejoin_strgen_timed(const char* words_path, uint64_t n, uint64_t seed_r, uint64_t seed_s, unsigned threads,
                                      int reps, uint64_t* out3) {
  const std::vector<std::string> words = hmj_strgen::load_words(words_path);
  if (words.empty()) return -1.0;
  KeyValVec r = hmj_strgen::create_strvec((int)n, words, seed_r), s = hmj_strgen::create_strvec((int)n, words, seed_s);
  double best = 1e30;
  for (int it = 0; it < reps; it++) {
    const auto t0 = std::chrono::steady_clock::now();
    HashMergeJoin<KeyValVec::iterator, KeyValVec::iterator> hmj(r.begin(), r.end(), s.begin(), s.end(), threads);
    // the timed region is the benchmark's own (hashjoin_bench.cc:126-133): construct, iterate, reduce to one sum
    uint64_t sum = 0;
    for (auto t : hmj) sum += *std::get<1>(t) + *std::get<2>(t);
    const volatile uint64_t keep = sum;
    (void)keep;
    const double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    if (sec < best) best = sec;
    // the check is not timed: a second walk over the same join for the count and the ordered FNV of the pairs
    uint64_t cnt = 0, sum2 = 0, fnv = 0xCBF29CE484222325ull;
    for (auto t : hmj) {
      const uint64_t w[2] = {*std::get<1>(t), *std::get<2>(t)};
      sum2 += w[0] + w[1];
      cnt++;
      for (int q = 0; q < 2; q++)
        for (int b = 0; b < 8; b++) {
          fnv ^= (w[q] >> (8 * b)) & 0xFF;
          fnv *= 0x100000001B3ull;
        }
    }
    if (out3) {
      out3[0] = cnt;
      out3[1] = sum2 == sum ? sum : ~0ull;  // (the two walks must agree)
      out3[2] = fnv;
    }
  }
  return best;
}

}  // extern "C"
