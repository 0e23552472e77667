// hashmergejoin_hip.hpp -- drop-in replacement for the reference's join operator
// (dryman/HashMergeJoin hashjoin.h:33-199) that runs on an MI355X through the C ABI of hmj.h.
//
//   #include "hashmergejoin_hip.hpp"      // instead of "hashjoin.h"
//   HashMergeJoin<It,It> hmj(r.begin(), r.end(), s.begin(), s.end(), threads);
//   for (auto t : hmj) sum += *std::get<1>(t) + *std::get<2>(t);   // hashjoin_bench.cc:126-133
//
// Same names, arguments and iteration surface as the reference class:
//   static_asserts on key / difference types ........ hashjoin.h:35-42
//   HashMergeJoin() = default ........................ hashjoin.h:55
//   HashMergeJoin(r_begin,r_end,s_begin,s_end,n=1) ... hashjoin.h:56-68
//   iterator: ++, ++(int), ==, !=, * ................. hashjoin.h:104-173
//   begin(), end(), clear() .......................... hashjoin.h:183-195
// operator* yields std::tuple<Key*,RValue*,SValue*>& whose pointers stay valid while the join
// object lives (the reference points into its sorted copies; this class points into its result
// columns).  Rows come in ascending key order, as the reference's do.
//
// Host code only: compile with any C++11 compiler and link -lhmj_hip.  Key = uint64_t with 8-byte
// trivially copyable payloads in contiguous storage (std::vector<std::pair<uint64_t,V>>, SURVEY.md
// D4) goes to the GPU as is.  Any other hashable key type (std::string, the reference's KeyValVec)
// is hashed on the host with std::hash<Key> -- as the reference does -- and joined on the GPU as
// {hash, row index} rows.  The join itself always runs on the GPU: there is no CPU join path.
// Errors (the reference has none: assert/UB) surface as std::runtime_error.
//
// Semantics note: relational equi-join.  Identical to the reference for relations whose keys are
// unique per relation (what its generator produces, strgen_test.cc:24-33); with duplicate keys
// this class yields the full cross product per key where the reference iterator yields a
// "staircase" and may drop tail matches.
#ifndef HASHMERGEJOIN_HIP_HPP
#define HASHMERGEJOIN_HIP_HPP 1

#include <algorithm>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <iterator>
#include <memory>
#include <stdexcept>
#include <string>
#include <thread>
#include <tuple>
#include <type_traits>
#include <utility>
#include <vector>

#include "hmj.h"

namespace hmj_detail {

// one executor context per calling thread (an hmj_ctx is not thread-safe), created on first use
struct ThreadCtx {
  hmj_ctx* ctx = nullptr;
  ~ThreadCtx() {
    if (ctx) hmj_destroy(ctx);
  }
};
inline hmj_ctx* thread_ctx() {
  static thread_local ThreadCtx tc;
  if (!tc.ctx) {
    int rc = hmj_create(&tc.ctx, -1);
    if (rc != HMJ_OK) throw std::runtime_error(std::string("hmj_create: ") + hmj_strerror(rc));
  }
  return tc.ctx;
}
inline void check(hmj_ctx* c, int rc, const char* what) {
  if (rc != HMJ_OK)
    throw std::runtime_error(std::string(what) + ": " + hmj_strerror(rc) + " (" + hmj_last_error(c) + ")");
}

// A relation the GPU path takes: std::pair<uint64_t, 8-byte trivially copyable payload> rows behind a random
// access iterator -- all the reference asks of its iterators is `begin + k` (radix_hash.h:375-388).
template <typename Iter>
struct is_hmj_relation_iter {
  typedef typename std::iterator_traits<Iter>::value_type item;
  static const bool value = sizeof(item) == 16 &&
                            std::is_same<typename item::first_type, std::uint64_t>::value &&
                            sizeof(typename item::second_type) == 8 &&
                            std::is_trivially_copyable<typename item::second_type>::value &&
                            std::is_base_of<std::random_access_iterator_tag,
                                            typename std::iterator_traits<Iter>::iterator_category>::value;
};

// Only a range known to be CONTIGUOUS has one address to hand to the C ABI: raw pointers and std::vector's
// iterators (C++11 has no contiguous_iterator_tag to ask).  Every other random access iterator -- std::deque's,
// a reverse_iterator, a strided view -- is legal for the reference and is copied row by row into a contiguous
// staging vector first (relation_rows below); it is never read as `addressof(*begin)` + n.
template <typename Iter>
struct is_contiguous_iter {
  typedef typename std::remove_cv<typename std::iterator_traits<Iter>::value_type>::type T;
  static const bool value = std::is_pointer<Iter>::value ||
                            std::is_same<Iter, typename std::vector<T>::iterator>::value ||
                            std::is_same<Iter, typename std::vector<T>::const_iterator>::value;
};

typedef std::vector<std::pair<std::uint64_t, std::uint64_t>> StagedRows;

template <typename T>
inline void check_pair_layout(const T& row) {  // std::pair<uint64_t,V> must be laid out {first, second}
  const char* b = reinterpret_cast<const char*>(std::addressof(row));
  if (reinterpret_cast<const char*>(std::addressof(row.first)) != b ||
      reinterpret_cast<const char*>(std::addressof(row.second)) - b != 8)
    throw std::runtime_error("HashMergeJoin: unexpected std::pair layout");
}
template <typename Iter>
inline const void* relation_rows(Iter begin, std::size_t n, StagedRows&, std::true_type /*contiguous*/) {
  if (!n) return nullptr;
  check_pair_layout(*begin);
  return static_cast<const void*>(std::addressof(*begin));
}
template <typename Iter>
inline const void* relation_rows(Iter begin, std::size_t n, StagedRows& stage, std::false_type) {
  if (!n) return nullptr;
  check_pair_layout(*begin);
  stage.resize(n);
  for (std::size_t i = 0; i < n; i++, ++begin) {
    stage[i].first = begin->first;
    std::memcpy(&stage[i].second, std::addressof(begin->second), 8);
  }
  return static_cast<const void*>(stage.data());
}
// the rows of [begin, begin + n) as one contiguous block of n x {u64 key, 8-byte payload}: the caller's own
// storage when the iterator type guarantees contiguity, else a copy held by `stage`
template <typename Iter>
inline const void* relation_rows(Iter begin, std::size_t n, StagedRows& stage) {
  return relation_rows(begin, n, stage, std::integral_constant<bool, is_contiguous_iter<Iter>::value>());
}

// Join two relations given as {64-bit hash, row index} rows on the GPU and keep the pairs whose KEYS are
// equal (eq(r_row, s_row)); inside a run of equal hashes the pairs are ordered by key (less(r_row_a,
// r_row_b)), as the reference's sort does (radix_hash.h:86-109 breaks hash ties on the key).  Fills the
// matching row indices in iteration order.
// what a key points at, if anything (std::string: its characters; inline up to 15 of them, on the heap beyond)
template <typename K>
inline void prefetch_key_data(const K&) {}
inline void prefetch_key_data(const std::string& k) { __builtin_prefetch(k.data()); }
// a per-thread buffer of 64-bit words that only grows and is never initialised (the {hash, row} rows of the string-key operator)
// hash_scratch(0, true) after use gives a buffer beyond 256 MiB back (a one-off join of 10^8 strings must not pin gigabytes
// in its thread for ever; the usual sizes stay, so that the next join finds its pages faulted in).
inline std::uint64_t* hash_scratch(std::size_t words, bool done = false) {
  thread_local std::unique_ptr<std::uint64_t[]> buf;
  thread_local std::size_t cap = 0;
  if (done) {
    if (cap > (std::size_t)32 << 20) {
      buf.reset();
      cap = 0;
    }
    return nullptr;
  }
  if (words > cap) {
    buf.reset();
    buf.reset(new std::uint64_t[words + words / 8 + 16]);
    cap = words + words / 8 + 16;
  }
  return buf.get();
}
// fn(begin, end) over [0, n) on up to `threads` host threads (the caller's thread takes the last share)
template <typename Fn>
inline void parallel_ranges(std::size_t n, unsigned threads, Fn fn) {
  if (threads > 64) threads = 64;
  if (threads < 2 || n < 65536) {
    fn((std::size_t)0, n);
    return;
  }
  const std::size_t per = (n + threads - 1) / threads;
  std::vector<std::thread> th;
  for (unsigned t = 0; t + 1 < threads; t++) {
    const std::size_t b = (std::size_t)t * per, e = b + per < n ? b + per : n;
    if (b < e) th.emplace_back(fn, b, e);
  }
  const std::size_t b = (std::size_t)(threads - 1) * per;
  if (b < n) fn(b, n);
  for (auto& x : th) x.join();
}

template <typename Eq, typename Less>
inline void join_hashed_rows(const std::vector<std::pair<std::uint64_t, std::uint64_t>>& hr,
                             const std::vector<std::pair<std::uint64_t, std::uint64_t>>& hs, unsigned num_threads,
                             Eq eq, Less less, std::vector<std::uint64_t>& ri, std::vector<std::uint64_t>& si) {
  hmj_ctx* c = thread_ctx();
  hmj_set_host_threads(c, num_threads > 16 ? 16 : (int)num_threads);
  hmj_result res;
  hmj_rows* rows = nullptr;
  check(c, hmj_join_u64_rows(c, hr.empty() ? nullptr : hr.data(), hr.size(), hs.empty() ? nullptr : hs.data(),
                             hs.size(), HMJ_MATERIALIZE | HMJ_ORDERED, &res, &rows), "hmj_join_u64_rows");
  std::shared_ptr<hmj_rows> guard(rows, hmj_rows_free);
  // rows are (hash, r index, s index) in ascending hash
  const std::size_t n = (std::size_t)res.n_matches;
  ri.clear();
  si.clear();
  // The usual case -- no two different keys share a 64-bit hash, so every pair the GPU found is a pair of equal keys and no
  // hash value occurs twice among the results -- is verified on all host threads (each pair costs two cache misses into the
  // caller's relations: 10^6 pairs took 80 ms on one thread, configs[0]); then the GPU's columns ARE the answer.
  bool simple = true;
  {
    std::vector<char> bad_flag(1, 0);
    char* bad = bad_flag.data();
    parallel_ranges(n, num_threads, [&, bad](std::size_t b, std::size_t e) {
      bool any = false;
      for (std::size_t k = b; k < e && !any; k++) {
        if (!eq(res.rval[k], res.sval[k])) any = true;                 // a hash collision between different keys
        if (k + 1 < n && res.key[k + 1] == res.key[k]) any = true;     // several result rows share a hash value
      }
      if (any) *bad = 1;  // (benign race: every writer stores the same value)
    });
    simple = *bad == 0;
  }
  if (simple) {
    ri.assign(res.rval, res.rval + n);
    si.assign(res.sval, res.sval + n);
    return;
  }
  ri.reserve(n);
  si.reserve(n);
  std::size_t i = 0;
  while (i < n) {
    std::size_t j = i + 1;
    while (j < n && res.key[j] == res.key[i]) j++;
    const std::size_t first = ri.size();
    for (std::size_t k = i; k < j; k++)
      if (eq(res.rval[k], res.sval[k])) {
        ri.push_back(res.rval[k]);
        si.push_back(res.sval[k]);
      }
    if (ri.size() - first > 1) {  // several rows share this hash: order them by key
      std::vector<std::pair<std::uint64_t, std::uint64_t>> grp;
      for (std::size_t k = first; k < ri.size(); k++) grp.emplace_back(ri[k], si[k]);
      std::stable_sort(grp.begin(), grp.end(),
                       [&](const std::pair<std::uint64_t, std::uint64_t>& a,
                           const std::pair<std::uint64_t, std::uint64_t>& b) { return less(a.first, b.first); });
      for (std::size_t k = 0; k < grp.size(); k++) {
        ri[first + k] = grp[k].first;
        si[first + k] = grp[k].second;
      }
    }
    i = j;
  }
}

// The same join for a caller that also wants something from every result pair (the string-key operator copies the
// payloads): in the usual case -- no collision, no repeated hash -- the GPU's columns ARE the answer, so they are kept where
// they are (pooled host memory, held by `rows`) instead of being copied into vectors, and visit(k, r_row, s_row) runs inside
// the verification pass, on the thread that has just pulled both rows into its cache (a second pass over 10^6 pairs misses
// the cache twice per pair again).  prep(n) is called once before the pass.  Otherwise the pairs end up in ri_own / si_own
// as join_hashed_rows leaves them and `visited` is false: the caller walks them itself.
struct HashedJoin {
  std::shared_ptr<hmj_rows> rows;
  const std::uint64_t* ri = nullptr;
  const std::uint64_t* si = nullptr;
  std::size_t n = 0;
  std::vector<std::uint64_t> ri_own, si_own;
  bool visited = false;
  void clear() {
    rows.reset();
    ri = si = nullptr;
    n = 0;
    ri_own.clear();
    si_own.clear();
    visited = false;
  }
};
// touch(stage, r_row, s_row) is called a few pairs AHEAD of eq / visit: the caller prefetches the two rows (stage 0, twelve
// pairs ahead) and then what the rows point at (stage 1, six ahead: the characters of a std::string beyond its 15 inline
// ones) -- every pair is two to four cache misses into the caller's relations, and a thread that waits for them one pair at
// a time leaves most of its memory parallelism unused.
template <typename Eq, typename Less, typename Prep, typename Visit, typename Touch>
inline void join_hashed_rows_visit(const std::uint64_t* hr, std::size_t nr, const std::uint64_t* hs, std::size_t ns,
                                   unsigned num_threads, Eq eq, Less less, Prep prep, Visit visit, Touch touch, HashedJoin& out) {
  // hr / hs: nr / ns rows of {hash, row index}, two 64-bit words each
  out.clear();
  const auto tstart = std::chrono::steady_clock::now();
  hmj_ctx* c = thread_ctx();
  hmj_set_host_threads(c, num_threads > 16 ? 16 : (int)num_threads);
  hmj_result res;
  hmj_rows* rows = nullptr;
  check(c, hmj_join_u64_rows(c, nr ? hr : nullptr, nr, ns ? hs : nullptr, ns, HMJ_MATERIALIZE | HMJ_ORDERED, &res, &rows),
        "hmj_join_u64_rows");
  std::shared_ptr<hmj_rows> guard(rows, hmj_rows_free);
  const std::size_t n = (std::size_t)res.n_matches;
  const bool times = std::getenv("HMJ_DROPIN_TIMES") != nullptr;
  const auto tj = std::chrono::steady_clock::now();
  prep(n);
  const auto tp = std::chrono::steady_clock::now();
  std::vector<char> bad_flag(1, 0);
  char* bad = bad_flag.data();
  parallel_ranges(n, num_threads, [&, bad](std::size_t b, std::size_t e) {
    constexpr std::size_t kAhead = 12, kNear = 6;  // stage 0: the rows themselves; stage 1: what they point at (a string's characters)
    for (std::size_t k = b; k < e && k < b + kAhead; k++) touch(0, res.rval[k], res.sval[k]);
    for (std::size_t k = b; k < e && k < b + kNear; k++) touch(1, res.rval[k], res.sval[k]);
    for (std::size_t k = b; k < e; k++) {
      if (k + kAhead < e) touch(0, res.rval[k + kAhead], res.sval[k + kAhead]);
      if (k + kNear < e) touch(1, res.rval[k + kNear], res.sval[k + kNear]);
      const std::uint64_t r = res.rval[k], q = res.sval[k];
      if (!eq(r, q) || (k + 1 < n && res.key[k + 1] == res.key[k])) {  // a collision between different keys / a repeated hash
        *bad = 1;  // (benign race: every writer stores the same value)
        return;
      }
      visit(k, r, q);
    }
  });
  if (times)
    std::fprintf(stderr, "[hmj drop-in] join of {hash,row} rows %.2f ms (from its start), result arrays %.2f ms, verify + visit %.2f ms\n",
                 std::chrono::duration<double, std::milli>(tj - tstart).count(), std::chrono::duration<double, std::milli>(tp - tj).count(),
                 std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tp).count());
  if (*bad == 0) {
    out.rows = guard;
    out.ri = res.rval;
    out.si = res.sval;
    out.n = n;
    out.visited = true;
    return;
  }
  // (rare: the general walk of join_hashed_rows over the rows already here)
  std::vector<std::uint64_t>& ri = out.ri_own;
  std::vector<std::uint64_t>& si = out.si_own;
  std::size_t i = 0;
  while (i < n) {
    std::size_t j = i + 1;
    while (j < n && res.key[j] == res.key[i]) j++;
    const std::size_t first = ri.size();
    for (std::size_t k = i; k < j; k++)
      if (eq(res.rval[k], res.sval[k])) {
        ri.push_back(res.rval[k]);
        si.push_back(res.sval[k]);
      }
    if (ri.size() - first > 1) {  // several rows share this hash: order them by key
      std::vector<std::pair<std::uint64_t, std::uint64_t>> grp;
      for (std::size_t k = first; k < ri.size(); k++) grp.emplace_back(ri[k], si[k]);
      std::stable_sort(grp.begin(), grp.end(),
                       [&](const std::pair<std::uint64_t, std::uint64_t>& a,
                           const std::pair<std::uint64_t, std::uint64_t>& b) { return less(a.first, b.first); });
      for (std::size_t k = 0; k < grp.size(); k++) {
        ri[first + k] = grp[k].first;
        si[first + k] = grp[k].second;
      }
    }
    i = j;
  }
  out.ri = ri.data();
  out.si = si.data();
  out.n = ri.size();
}
}  // namespace hmj_detail


// Primary template.  Native == true : uint64_t keys with 8-byte payloads in contiguous storage go to the
//                                       GPU as they are (the path BASELINE.json names).
//                    Native == false: any other key type with std::hash / == / < (e.g. the reference's
//                                       KeyValVec of std::string keys, hashjoin.h:29): see below.
template <typename RIter, typename SIter,
          bool Native = hmj_detail::is_hmj_relation_iter<RIter>::value && hmj_detail::is_hmj_relation_iter<SIter>::value>
class HashMergeJoin;

template <typename RIter, typename SIter>
class HashMergeJoin<RIter, SIter, true> {
  static_assert(std::is_same<typename std::iterator_traits<RIter>::value_type::first_type,
                             typename std::iterator_traits<SIter>::value_type::first_type>::value,
                "RIter and SIter key type must be the same");
  static_assert(std::is_same<typename std::iterator_traits<RIter>::difference_type, typename std::iterator_traits<SIter>::difference_type>::value,
                "RIter and SIter difference type must be the same");

  typedef typename std::iterator_traits<RIter>::difference_type distance_type;
  typedef typename std::iterator_traits<RIter>::value_type::first_type Key;
  typedef typename std::iterator_traits<RIter>::value_type::second_type RValue;
  typedef typename std::iterator_traits<SIter>::value_type::second_type SValue;

 public:
  HashMergeJoin() = default;
  // num_threads (hashjoin.h:58) sets the number of host threads that stage the relations for the
  // PCIe copy; the join itself runs on the GPU.
  HashMergeJoin(RIter r_begin, RIter r_end, SIter s_begin, SIter s_end, unsigned int num_threads = 1) {
    const distance_type r_size = std::distance(r_begin, r_end), s_size = std::distance(s_begin, s_end);
    hmj_detail::StagedRows r_stage, s_stage;  // used only by iterators that are not known to be contiguous
    const void* r_ptr = hmj_detail::relation_rows(r_begin, (std::size_t)r_size, r_stage);
    const void* s_ptr = hmj_detail::relation_rows(s_begin, (std::size_t)s_size, s_stage);
    hmj_ctx* c = hmj_detail::thread_ctx();
    hmj_set_host_threads(c, num_threads > 16 ? 16 : (int)num_threads);
    hmj_result res;
    hmj_rows* rows = nullptr;
    hmj_detail::check(c, hmj_join_u64_rows(c, r_ptr, (uint64_t)r_size, s_ptr, (uint64_t)s_size,
                                           HMJ_MATERIALIZE | HMJ_ORDERED, &res, &rows), "hmj_join_u64_rows");
    _rows = std::shared_ptr<hmj_rows>(rows, hmj_rows_free);  // the result columns (host memory)
    _n = (std::size_t)res.n_matches;
    _key = const_cast<Key*>(res.key);
    _rval = reinterpret_cast<RValue*>(const_cast<uint64_t*>(res.rval));
    _sval = reinterpret_cast<SValue*>(const_cast<uint64_t*>(res.sval));
  }

  class iterator : public std::iterator<std::input_iterator_tag, std::tuple<Key*, RValue*, SValue*>> {
   public:
    iterator(HashMergeJoin* owner, std::size_t pos) : _owner(owner), _pos(pos) {}
    iterator& operator++() {
      ++_pos;
      return *this;
    }
    iterator operator++(int) {
      iterator retval = *this;
      ++(*this);
      return retval;
    }
    bool operator==(iterator other) const { return _pos == other._pos; }
    bool operator!=(iterator other) const { return _pos != other._pos; }
    std::tuple<Key*, RValue*, SValue*>& operator*() {
      tmp_val = std::make_tuple(_owner->_key + _pos, _owner->_rval + _pos, _owner->_sval + _pos);
      return tmp_val;
    }

   protected:
    HashMergeJoin* _owner;
    std::size_t _pos;
    std::tuple<Key*, RValue*, SValue*> tmp_val;
  };

  iterator begin() { return iterator(this, 0); }
  iterator end() { return iterator(this, _n); }
  void clear() {
    _rows.reset();
    _n = 0;
    _key = nullptr;
    _rval = nullptr;
    _sval = nullptr;
  }
  // not in the reference: number of result rows
  std::size_t size() const { return _n; }

 protected:
  // Result columns live in host memory owned by _rows (shared by copies of this object, as
  // the reference's copies share nothing but are equally valid while they live).
  std::shared_ptr<hmj_rows> _rows;
  std::size_t _n = 0;
  Key* _key = nullptr;
  RValue* _rval = nullptr;
  SValue* _sval = nullptr;
};

// ---------------------------------------------------------------------------------------------------
// Keys of any hashable type (the reference's own benchmark joins std::string keys,
// hashjoin_bench.cc:109-143).  The reference sorts (hash, key, value) tuples by std::hash<Key>
// (hashjoin.h:65-67 -> radix_hash.h:314) and merges on (hash, key).  Here the host computes the same
// std::hash<Key> once per row (the reference computes it twice, radix_hash.h:314,340) with
// `num_threads` threads, the GPU joins the 16-byte rows {hash, row index} in hash order, and the host
// drops the (astronomically rare) pairs whose 64-bit hashes collide on different keys.  Iteration order
// is the reference's: ascending hash, then key.  operator* points at the KEY inside the caller's relation
// (the reference points into its sorted copies), so the relations must outlive the join; the payload
// pointers point at copies the join object keeps in iteration order (as the reference's do).
// ---------------------------------------------------------------------------------------------------
template <typename RIter, typename SIter>
class HashMergeJoin<RIter, SIter, false> {
  static_assert(std::is_same<typename std::iterator_traits<RIter>::value_type::first_type,
                             typename std::iterator_traits<SIter>::value_type::first_type>::value,
                "RIter and SIter key type must be the same");
  static_assert(std::is_same<typename std::iterator_traits<RIter>::difference_type, typename std::iterator_traits<SIter>::difference_type>::value,
                "RIter and SIter difference type must be the same");
  static_assert(std::is_base_of<std::random_access_iterator_tag,
                                typename std::iterator_traits<RIter>::iterator_category>::value &&
                    std::is_base_of<std::random_access_iterator_tag,
                                    typename std::iterator_traits<SIter>::iterator_category>::value,
                "random access iterators are required");

  typedef typename std::iterator_traits<RIter>::difference_type distance_type;
  typedef typename std::iterator_traits<RIter>::value_type::first_type Key;
  typedef typename std::iterator_traits<RIter>::value_type::second_type RValue;
  typedef typename std::iterator_traits<SIter>::value_type::second_type SValue;

 public:
  HashMergeJoin() = default;
  HashMergeJoin(RIter r_begin, RIter r_end, SIter s_begin, SIter s_end, unsigned int num_threads = 1)
      : _r(r_begin), _s(s_begin) {
    const std::size_t nr = (std::size_t)std::distance(r_begin, r_end), ns = (std::size_t)std::distance(s_begin, s_end);
    const bool times = std::getenv("HMJ_DROPIN_TIMES") != nullptr;  // (developer aid: the ctor's phases on stderr)
    const auto t0 = std::chrono::steady_clock::now();
    // {hash, row index} rows of both relations in one scratch buffer that this thread keeps from join to join and that is
    // never zero-filled (10^6 + 10^6 rows: two fresh std::vectors cost 7 of the 11 ms this phase took -- 32 MiB of page
    // faults and zeroes on one thread --, the hashing itself 3), filled by ONE team of threads
    std::uint64_t* const hr = hmj_detail::hash_scratch(2 * (nr + ns));
    std::uint64_t* const hs = hr + 2 * nr;
    hmj_detail::parallel_ranges(nr + ns, num_threads, [&](std::size_t b, std::size_t e) {
      std::hash<Key> h;
      for (std::size_t i = b; i < e; i++) {
        // (the rows are read in order, but a long string's characters live wherever the allocator put them)
        if (i + 16 < e) hmj_detail::prefetch_key_data(i + 16 < nr ? r_begin[i + 16].first : s_begin[i + 16 - nr].first);
        if (i < nr) {
          hr[2 * i] = (std::uint64_t)h(r_begin[i].first);
          hr[2 * i + 1] = (std::uint64_t)i;
        } else {
          hs[2 * (i - nr)] = (std::uint64_t)h(s_begin[i - nr].first);
          hs[2 * (i - nr) + 1] = (std::uint64_t)(i - nr);
        }
      }
    });
    const auto t1 = std::chrono::steady_clock::now();
    // The payloads of the result rows are copied into the join object, in iteration order -- the reference's iterator
    // walks its own sorted copies too (hashjoin.h:168-173) -- so that iterating does not miss the cache twice per row in
    // the caller's relations (10^6 rows: 126 ms of a 239 ms join, configs[0]).  The copy happens inside the pass that
    // verifies the pairs' keys, which has just pulled both rows into the cache, on all host threads.
    RValue* rv = nullptr;
    SValue* sv = nullptr;
    auto alloc = [&](std::size_t n) {
      _rv = std::shared_ptr<RValue>(new RValue[n ? n : 1], std::default_delete<RValue[]>());
      _sv = std::shared_ptr<SValue>(new SValue[n ? n : 1], std::default_delete<SValue[]>());
      rv = _rv.get();
      sv = _sv.get();
    };
    hmj_detail::join_hashed_rows_visit(
        hr, nr, hs, ns, num_threads,
        [&](std::uint64_t r, std::uint64_t q) { return r_begin[r].first == s_begin[q].first; },
        [&](std::uint64_t x, std::uint64_t y) { return r_begin[x].first < r_begin[y].first; }, alloc,
        [&](std::size_t k, std::uint64_t r, std::uint64_t q) {
          rv[k] = r_begin[r].second;
          sv[k] = s_begin[q].second;
        },
        [&](int stage, std::uint64_t r, std::uint64_t q) {
          if (stage == 0) {
            __builtin_prefetch(&r_begin[r]);
            __builtin_prefetch(&s_begin[q]);
          } else {
            hmj_detail::prefetch_key_data(r_begin[r].first);
            hmj_detail::prefetch_key_data(s_begin[q].first);
          }
        },
        _j);
    if (times)
      std::fprintf(stderr, "[hmj drop-in] %zu x %zu rows: hash %.2f ms, GPU join + verify + payloads %.2f ms\n", nr, ns,
                   std::chrono::duration<double, std::milli>(t1 - t0).count(),
                   std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t1).count());
    (void)hmj_detail::hash_scratch(0, true);
    if (!_j.visited) {  // (colliding or repeated hashes: the pairs were walked one by one; their payloads now)
      alloc(_j.n);
      hmj_detail::parallel_ranges(_j.n, num_threads, [&](std::size_t b, std::size_t e) {
        for (std::size_t k = b; k < e; k++) {
          rv[k] = r_begin[_j.ri[k]].second;
          sv[k] = s_begin[_j.si[k]].second;
        }
      });
    }
  }

  class iterator : public std::iterator<std::input_iterator_tag, std::tuple<Key*, RValue*, SValue*>> {
   public:
    iterator(HashMergeJoin* owner, std::size_t pos) : _owner(owner), _pos(pos) {}
    iterator& operator++() {
      ++_pos;
      return *this;
    }
    iterator operator++(int) {
      iterator retval = *this;
      ++(*this);
      return retval;
    }
    bool operator==(iterator other) const { return _pos == other._pos; }
    bool operator!=(iterator other) const { return _pos != other._pos; }
    std::tuple<Key*, RValue*, SValue*>& operator*() {
      // (the key's ADDRESS in the caller's relation -- nothing of the row is read unless the caller dereferences it)
      const typename std::iterator_traits<RIter>::value_type& rr = _owner->_r[_owner->_j.ri[_pos]];
      tmp_val = std::make_tuple(const_cast<Key*>(&rr.first), _owner->_rv.get() + _pos, _owner->_sv.get() + _pos);
      return tmp_val;
    }

   protected:
    HashMergeJoin* _owner;
    std::size_t _pos;
    std::tuple<Key*, RValue*, SValue*> tmp_val;
  };

  iterator begin() { return iterator(this, 0); }
  iterator end() { return iterator(this, _j.n); }
  void clear() {
    _j.clear();
    _rv.reset();
    _sv.reset();
  }
  std::size_t size() const { return _j.n; }

 protected:
  RIter _r;
  SIter _s;
  hmj_detail::HashedJoin _j;    // matching row indices into the caller's relations (the GPU's result columns, kept where they are)
  std::shared_ptr<RValue> _rv;  // the result rows' payloads, in iteration order (arrays; copies of the join share them)
  std::shared_ptr<SValue> _sv;
};

// ---------------------------------------------------------------------------------------------------
// HashMergeJoin2 (hashjoin.h:201-363): the relations arrive PRE-HASHED as std::tuple<size_t hash, Key,
// Value> rows.  The reference sorts both buffers in place by the hash (radix_inplace_par,
// radix_hash.h:589-654; keys only break ties inside its insertion sort of tiny buckets) and merges them on
// (hash, key); here both buffers are sorted by hash (stable), the GPU joins the {hash, row index} rows, the host keeps
// the pairs whose keys are equal, and iteration runs in the same ascending (hash, key) order.
// operator* points into the caller's buffers, as the reference's does -- and, as in the reference
// (hashjoin.h:234-235), the ctor leaves BOTH caller buffers sorted by hash: tuples of 8-byte trivially copyable
// elements in contiguous storage are sorted on the GPU as they are (hmj_sort_rows_by_u64_host); any other
// tuple type is permuted on the host into the order the GPU computes (hmj_argsort_u64_host).  The sort is
// stable where the reference's is not: rows with equal hashes keep their input order.
// ---------------------------------------------------------------------------------------------------
template <typename RIter, typename SIter>
class HashMergeJoin2 {
  static_assert(std::is_same<typename std::tuple_element<1, typename std::iterator_traits<RIter>::value_type>::type,
                             typename std::tuple_element<1, typename std::iterator_traits<SIter>::value_type>::type>::value,
                "RIter and SIter key type must be the same");
  static_assert(std::is_same<typename std::iterator_traits<RIter>::difference_type, typename std::iterator_traits<SIter>::difference_type>::value,
                "RIter and SIter difference type must be the same");
  static_assert(std::is_base_of<std::random_access_iterator_tag,
                                typename std::iterator_traits<RIter>::iterator_category>::value &&
                    std::is_base_of<std::random_access_iterator_tag,
                                    typename std::iterator_traits<SIter>::iterator_category>::value,
                "random access iterators are required");

  typedef typename std::tuple_element<1, typename std::iterator_traits<RIter>::value_type>::type Key;
  typedef typename std::tuple_element<2, typename std::iterator_traits<RIter>::value_type>::type RValue;
  typedef typename std::tuple_element<2, typename std::iterator_traits<SIter>::value_type>::type SValue;

  template <typename Iter>
  static void hash_column(Iter begin, std::size_t n, std::vector<std::pair<std::uint64_t, std::uint64_t>>& out) {
    out.resize(n);
    for (std::size_t i = 0; i < n; i++) out[i] = std::make_pair((std::uint64_t)std::get<0>(begin[i]), (std::uint64_t)i);
  }

  // radix_inplace_par's visible effect (radix_hash.h:589-654): the caller's buffer ends up sorted by hash (stable: equal
  // hashes keep their input order; the reference's swap chains do not promise any)
  template <typename Iter>
  static void sort_in_place(Iter begin, std::size_t n) {
    typedef typename std::iterator_traits<Iter>::value_type T;
    typedef typename std::tuple_element<0, T>::type H;
    typedef typename std::tuple_element<1, T>::type K;
    typedef typename std::tuple_element<2, T>::type V;
    if (n < 2) return;
    hmj_ctx* c = hmj_detail::thread_ctx();
    const bool contiguous = std::is_pointer<Iter>::value || std::is_same<Iter, typename std::vector<T>::iterator>::value;
    // Rows may go through the device as raw bytes only when that is what copying a T means: every element
    // trivially copyable, and the tuple nothing but its elements -- no padding, no over-alignment (std::tuple itself
    // is never std::is_trivially_copyable in libstdc++: its assignment operators are user-provided; a tuple from
    // another standard library with a different layout still passes or fails this test on its own merits, and the
    // hash element's offset is taken from the object, not assumed).  Anything else: argsort + moves on the host.
    const bool plain = std::is_trivially_copyable<H>::value && std::is_trivially_copyable<K>::value &&
                       std::is_trivially_copyable<V>::value && sizeof(H) == 8 && sizeof(T) % 8 == 0 && sizeof(T) <= 64 &&
                       sizeof(T) == sizeof(H) + sizeof(K) + sizeof(V) && alignof(T) <= 8;
    if (contiguous && plain) {  // (only a contiguous range has ONE address to hand over)
      T* first = std::addressof(*begin);
      const std::size_t hash_off = (std::size_t)(reinterpret_cast<const char*>(&std::get<0>(*first)) - reinterpret_cast<const char*>(first));
      hmj_detail::check(c, hmj_sort_rows_by_u64_host(c, first, (uint64_t)n, (uint32_t)sizeof(T), (uint32_t)hash_off), "hmj_sort_rows_by_u64_host");
      return;
    }
    std::vector<std::uint64_t> h(n);
    for (std::size_t i = 0; i < n; i++) h[i] = (std::uint64_t)std::get<0>(begin[i]);
    std::vector<std::uint32_t> perm(n);
    hmj_detail::check(c, hmj_argsort_u64_host(c, h.data(), (uint64_t)n, 8, perm.data()), "hmj_argsort_u64_host");
    std::vector<T> tmp;
    tmp.reserve(n);
    for (std::size_t i = 0; i < n; i++) tmp.push_back(std::move(begin[perm[i]]));
    for (std::size_t i = 0; i < n; i++) begin[i] = std::move(tmp[i]);
  }

 public:
  HashMergeJoin2() = default;
  HashMergeJoin2(RIter r_begin, RIter r_end, SIter s_begin, SIter s_end, unsigned int num_threads = 1)
      : _r(r_begin), _s(s_begin) {
    sort_in_place(r_begin, (std::size_t)std::distance(r_begin, r_end));  // hashjoin.h:234
    sort_in_place(s_begin, (std::size_t)std::distance(s_begin, s_end));  // hashjoin.h:235
    std::vector<std::pair<std::uint64_t, std::uint64_t>> hr, hs;
    hash_column(r_begin, (std::size_t)std::distance(r_begin, r_end), hr);
    hash_column(s_begin, (std::size_t)std::distance(s_begin, s_end), hs);
    hmj_detail::join_hashed_rows(
        hr, hs, num_threads,
        [&](std::uint64_t r, std::uint64_t q) { return std::get<1>(r_begin[r]) == std::get<1>(s_begin[q]); },
        [&](std::uint64_t x, std::uint64_t y) { return std::get<1>(r_begin[x]) < std::get<1>(r_begin[y]); }, _ri,
        _si);
  }

  class iterator : public std::iterator<std::input_iterator_tag, std::tuple<Key*, RValue*, SValue*>> {
   public:
    iterator(HashMergeJoin2* owner, std::size_t pos) : _owner(owner), _pos(pos) {}
    iterator& operator++() {
      ++_pos;
      return *this;
    }
    iterator operator++(int) {
      iterator retval = *this;
      ++(*this);
      return retval;
    }
    bool operator==(iterator other) const { return _pos == other._pos; }
    bool operator!=(iterator other) const { return _pos != other._pos; }
    std::tuple<Key*, RValue*, SValue*>& operator*() {
      typename std::iterator_traits<RIter>::value_type& rr = _owner->_r[_owner->_ri[_pos]];
      typename std::iterator_traits<SIter>::value_type& ss = _owner->_s[_owner->_si[_pos]];
      tmp_val = std::make_tuple(&std::get<1>(rr), &std::get<2>(rr), &std::get<2>(ss));
      return tmp_val;
    }

   protected:
    HashMergeJoin2* _owner;
    std::size_t _pos;
    std::tuple<Key*, RValue*, SValue*> tmp_val;
  };

  iterator begin() { return iterator(this, 0); }
  iterator end() { return iterator(this, _ri.size()); }
  void clear() {
    _ri.clear();
    _si.clear();
  }
  std::size_t size() const { return _ri.size(); }

 protected:
  RIter _r;
  SIter _s;
  std::vector<std::uint64_t> _ri, _si;  // matching row indices into the caller's buffers
};

// Convenience spelled the way BASELINE.json's north_star names the entry point.
template <typename RIter, typename SIter>
HashMergeJoin<RIter, SIter> join(RIter r_begin, RIter r_end, SIter s_begin, SIter s_end,
                                 unsigned int num_threads = 1) {
  return HashMergeJoin<RIter, SIter>(r_begin, r_end, s_begin, s_end, num_threads);
}

// The reduction the reference's benchmark performs over a join (hashjoin_bench.cc:131-133),
// without materialising rows: returns sum(rval + sval) and optionally the match count.
template <typename RIter, typename SIter>
std::uint64_t hash_merge_join_sum(RIter r_begin, RIter r_end, SIter s_begin, SIter s_end,
                                  std::uint64_t* n_matches = nullptr) {
  static_assert(hmj_detail::is_hmj_relation_iter<RIter>::value && hmj_detail::is_hmj_relation_iter<SIter>::value,
                "the MI355X executor joins std::pair<uint64_t, 8-byte payload> relations");
  const auto r_size = std::distance(r_begin, r_end), s_size = std::distance(s_begin, s_end);
  hmj_detail::StagedRows r_stage, s_stage;
  const void* r_ptr = hmj_detail::relation_rows(r_begin, (std::size_t)r_size, r_stage);
  const void* s_ptr = hmj_detail::relation_rows(s_begin, (std::size_t)s_size, s_stage);
  hmj_ctx* c = hmj_detail::thread_ctx();
  hmj_result res;
  hmj_detail::check(c, hmj_join_u64(c, r_ptr, (uint64_t)r_size, s_ptr, (uint64_t)s_size, 0, &res), "hmj_join_u64");
  if (n_matches) *n_matches = res.n_matches;
  return res.sum_r + res.sum_s;
}

#endif
