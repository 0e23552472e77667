/*
 * TEST INFRASTRUCTURE ONLY -- see hmj_oracle.h.  Plain-C restatement of the reference's
 * radix sort / partition / build / probe / merge-iterate algorithms,
 * written from the behaviour of the cited reference lines, not copied from them.
 * PARITY PINNED against oracle/_ref (compiled reference) and tests/golden/.
 */
#define _GNU_SOURCE
#include "hmj_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ======================================================================================
 * Synthetic relations and checksums
 * ==================================================================================== */

uint64_t orc_mix64(uint64_t x) {
  x ^= x >> 30;
  x *= 0xBF58476D1CE4E5B9ull;
  x ^= x >> 27;
  x *= 0x94D049BB133111EBull;
  x ^= x >> 31;
  return x;
}

uint64_t orc_unmix64(uint64_t x) {
  x ^= (x >> 31) ^ (x >> 62);
  x *= 0x319642B2D24D8EC3ull; /* inverse of 0x94D049BB133111EB mod 2^64 */
  x ^= (x >> 27) ^ (x >> 54);
  x *= 0x96DE1B173F119089ull; /* inverse of 0xBF58476D1CE4E5B9 mod 2^64 */
  x ^= (x >> 30) ^ (x >> 60);
  return x;
}

void orc_gen_build(uint64_t* aos, uint64_t n, uint64_t start, uint64_t seed) {
  for (uint64_t k = 0; k < n; k++) {
    uint64_t i = start + k;
    aos[2 * k] = orc_mix64(i + seed);
    aos[2 * k + 1] = i;
  }
}

void orc_gen_probe(uint64_t* aos, uint64_t n, uint64_t start, uint64_t n_build, uint64_t seed,
                   uint64_t miss_mod) {
  for (uint64_t k = 0; k < n; k++) {
    uint64_t j = start + k;
    uint64_t idx = n_build ? (ORC_PI_A * j + ORC_PI_B) % n_build : j;
    if (miss_mod && (j % miss_mod) == 0) idx += n_build;
    aos[2 * k] = orc_mix64(idx + seed);
    aos[2 * k + 1] = j ^ ORC_VAL_XOR;
  }
}

void orc_gen_from_cdf(uint64_t* aos, uint64_t n, uint64_t start, const uint64_t* thr,
                      uint64_t domain, uint64_t seed, uint64_t zseed) {
  for (uint64_t k = 0; k < n; k++) {
    uint64_t i = start + k;
    uint64_t u = orc_mix64(i ^ zseed);
    uint64_t lo = 0, hi = domain; /* first rank with thr[rank] >= u, clamped to domain-1 */
    while (lo < hi) {
      uint64_t mid = lo + (hi - lo) / 2;
      if (thr[mid] < u)
        lo = mid + 1;
      else
        hi = mid;
    }
    if (lo >= domain) lo = domain - 1;
    aos[2 * k] = orc_mix64(lo + seed);
    aos[2 * k + 1] = i;
  }
}

void orc_gen_uniform_domain(uint64_t* aos, uint64_t n, uint64_t start, uint64_t domain,
                            uint64_t seed, uint64_t zseed) {
  for (uint64_t k = 0; k < n; k++) {
    uint64_t j = start + k;
    aos[2 * k] = orc_mix64((orc_mix64(j ^ zseed) % domain) + seed);
    aos[2 * k + 1] = j ^ ORC_VAL_XOR;
  }
}

uint64_t orc_tmix(uint64_t key, uint64_t rval, uint64_t sval) {
  uint64_t t = orc_mix64(key);
  t = orc_mix64(t ^ rval);
  t = orc_mix64(t + sval);
  return t;
}

void orc_checks_of_triples(const uint64_t* t, uint64_t n, orc_checks* out) {
  orc_checks c = {0, 0, 0, 0, 0};
  for (uint64_t i = 0; i < n; i++) {
    uint64_t m = orc_tmix(t[3 * i], t[3 * i + 1], t[3 * i + 2]);
    c.n_matches++;
    c.sum_r += t[3 * i + 1];
    c.sum_s += t[3 * i + 2];
    c.xor_fold ^= m;
    c.mix_sum += m;
  }
  *out = c;
}

uint64_t orc_fnv1a_triples(const uint64_t* t, uint64_t n) {
  uint64_t h = 0xCBF29CE484222325ull;
  for (uint64_t i = 0; i < 3 * n; i++) {
    uint64_t w = t[i];
    for (int b = 0; b < 8; b++) {
      h ^= (w >> (8 * b)) & 0xFF;
      h *= 0x100000001B3ull;
    }
  }
  return h;
}

/* ======================================================================================
 * a2 -- radix_hash.h:38-57.  Same double arithmetic, same loop bounds.
 * ==================================================================================== */
int orc_optimal_partition(uint64_t input_num) {
  double best = 1.0;
  int pick = 0;
  for (int k = 6; k < 15; k++) {
    double lk = log((double)input_num) / log((double)(1 << k));
    double up = ceil(lk), down = floor(lk);
    double d = (lk - down < up - lk) ? lk - down : up - lk;
    if (input_num < (1ull << k)) return k;
    if (d <= best) {
      pick = k;
      best = d;
    }
  }
  return pick;
}

/* ======================================================================================
 * Generic element helpers.  An element is W consecutive u64 words; word 0 is the radix
 * source (hash for W==3, key for W==2).  W==3 breaks insertion-sort ties on word 1 (key),
 * as radix_hash.h:86-109 does; W==2 compares word 0 only (radix_sort.h:35-51).
 * ==================================================================================== */
static inline void el_copy(uint64_t* d, const uint64_t* s, int W) {
  for (int k = 0; k < W; k++) d[k] = s[k];
}
static inline void el_swap(uint64_t* a, uint64_t* b, int W) {
  for (int k = 0; k < W; k++) {
    uint64_t t = a[k];
    a[k] = b[k];
    b[k] = t;
  }
}

/* radix_hash.h:86-120 / radix_sort.h:35-62 */
static void insertion_sort(uint64_t* a, uint64_t begin, uint64_t end, int W) {
  for (uint64_t idx = begin + 1; idx < end; idx++) {
    uint64_t i = idx;
    while (i > begin) {
      uint64_t* cur = a + i * W;
      uint64_t* prv = a + (i - 1) * W;
      if (cur[0] > prv[0]) break;
      if (cur[0] < prv[0] || (W == 3 && cur[1] < prv[1])) {
        el_swap(cur, prv, W);
        i--;
        continue;
      }
      break;
    }
  }
}

/* In-place counting permutation of a[begin,end) on digit(word0) -- the cycle-leader loop of
 * radix_hash.h:165-185 / :254-275 / radix_sort.h:112-131 / :201-222.  first[]/last[] are the
 * bucket cursors/ends (P entries each), already initialised from the histogram. */
static void cycle_permute(uint64_t* a, int W, int P, uint64_t* first, const uint64_t* last,
                          uint64_t mask, int shift) {
  uint64_t tmp[3];
  int iter = 0;
  while (iter < P) {
    uint64_t i = first[iter];
    if (i >= last[iter]) {
      iter++;
      continue;
    }
    int d = (int)((a[i * W] & mask) >> shift);
    if (d == iter) {
      first[iter]++;
      continue;
    }
    el_copy(tmp, a + i * W, W);
    uint64_t j;
    do {
      d = (int)((tmp[0] & mask) >> shift);
      j = first[d]++;
      el_swap(a + j * W, tmp, W);
    } while (j > i);
  }
}

/* Pass 2+: radix_hash.h:122-292 (bf6_helper_p/_s) and radix_sort.h:64-238 (rs1_helper_p/_s).
 * Both helpers apply the same per-bucket rule; _p only distributes the outermost buckets over
 * threads, which does not change the result, so one recursive routine restates both. */
static void msd_refine(uint64_t* a, int W, uint64_t begin, uint64_t end, int mask_bits,
                       int bits) {
  int P = 1 << bits;
  uint64_t size = end - begin;
  if (size < 2) return;
  if (size < (uint64_t)(1 << (bits / 2))) {
    insertion_sort(a, begin, end, W);
    return;
  }
  uint64_t mask = (1ull << mask_bits) - 1ull;
  int shift = mask_bits < bits ? 0 : mask_bits - bits;
  uint64_t* cnt = (uint64_t*)calloc((size_t)P, sizeof(uint64_t));
  uint64_t* first = (uint64_t*)malloc((size_t)P * sizeof(uint64_t));
  uint64_t* last = (uint64_t*)calloc((size_t)P, sizeof(uint64_t));
  for (uint64_t i = begin; i < end; i++) cnt[(a[i * W] & mask) >> shift]++;
  uint64_t run = begin;
  for (int p = 0; p < P; p++) {
    first[p] = run;
    run += cnt[p];
    last[p] = run;
  }
  cycle_permute(a, W, P, first, last, mask, shift);
  int next_bits = mask_bits - bits;
  if (next_bits > 0) {
    uint64_t b = begin;
    for (int p = 0; p < P; p++) {
      msd_refine(a, W, b, last[p], next_bits, bits);
      b = last[p];
    }
  }
  free(cnt);
  free(first);
  free(last);
}

/* Pass 1: per-thread histogram, partition-major/thread-minor exclusive scan, stable scatter.
 * radix_hash.h:313-345 (W_out==3 writes hash,key,val) and radix_sort.h:418-449 (W_out==2).
 * Thread t owns rows [t*(n/T), (t+1)*(n/T)), the last thread also the remainder
 * (radix_hash.h:367,375-388).  Threads are replayed one after another: each owns private
 * counter rows, so the result equals any interleaving.  bucket_end[P] receives the ends. */
static void pass1_scatter(const uint64_t* aos, uint64_t n, int T, int shift, uint64_t digit_mask,
                          int P, int W_out, uint64_t* out, uint64_t* bucket_end) {
  if (T < 1) T = 1;
  uint64_t* cnt = (uint64_t*)calloc((size_t)P * (size_t)T, sizeof(uint64_t));
  uint64_t per = n / (uint64_t)T;
  for (int t = 0; t < T; t++) {
    uint64_t b = (uint64_t)t * per, e = (t == T - 1) ? n : b + per;
    for (uint64_t i = b; i < e; i++) cnt[(size_t)t * P + ((aos[2 * i] >> shift) & digit_mask)]++;
  }
  uint64_t run = 0;
  for (int p = 0; p < P; p++) {
    for (int t = 0; t < T; t++) {
      uint64_t c = cnt[(size_t)t * P + p];
      cnt[(size_t)t * P + p] = run;
      run += c;
    }
    bucket_end[p] = run;
  }
  for (int t = 0; t < T; t++) {
    uint64_t b = (uint64_t)t * per, e = (t == T - 1) ? n : b + per;
    for (uint64_t i = b; i < e; i++) {
      uint64_t key = aos[2 * i], val = aos[2 * i + 1];
      uint64_t dst = cnt[(size_t)t * P + ((key >> shift) & digit_mask)]++;
      if (W_out == 3) {
        out[3 * dst] = key; /* hash == key: std::hash<u64> is the identity */
        out[3 * dst + 1] = key;
        out[3 * dst + 2] = val;
      } else {
        out[2 * dst] = key;
        out[2 * dst + 1] = val;
      }
    }
  }
  free(cnt);
}

static void sort_non_inplace(const uint64_t* aos, uint64_t n, int threads, int bits, int W,
                             uint64_t* out) {
  if (bits < 0) bits = orc_optimal_partition(n);
  int P = 1 << bits;
  uint64_t* ends = (uint64_t*)malloc((size_t)P * sizeof(uint64_t));
  pass1_scatter(aos, n, threads, 64 - bits, ~0ull, P, W, out, ends);
  uint64_t b = 0;
  for (int p = 0; p < P; p++) { /* radix_hash.h:393-405 */
    msd_refine(out, W, b, ends[p], 64 - bits, bits);
    b = ends[p];
  }
  free(ends);
}

void orc_radix_non_inplace_par(const uint64_t* aos, uint64_t n, int threads, int bits,
                               uint64_t* out_hkv) {
  sort_non_inplace(aos, n, threads, bits, 3, out_hkv);
}

void orc_radix_int_non_inplace(const uint64_t* aos, uint64_t n, int threads, int bits,
                               uint64_t* out_aos) {
  sort_non_inplace(aos, n, threads, bits, 2, out_aos);
}

void orc_stable_partition(const uint64_t* aos, uint64_t n, int threads, int shift, int bits,
                          uint64_t* out_aos, uint64_t* offsets) {
  int P = 1 << bits;
  offsets[0] = 0;
  pass1_scatter(aos, n, threads, shift, (uint64_t)P - 1, P, 2, out_aos, offsets + 1);
}

/* radix_hash.h:425-483: top-level in-place counting permutation, then pass 2. */
void orc_radix_inplace_seq(uint64_t* hkv, uint64_t n, int bits) {
  if (bits < 0) bits = orc_optimal_partition(n);
  int P = 1 << bits, shift = 64 - bits;
  uint64_t* cnt = (uint64_t*)calloc((size_t)P, sizeof(uint64_t));
  uint64_t* first = (uint64_t*)malloc((size_t)P * sizeof(uint64_t));
  uint64_t* last = (uint64_t*)calloc((size_t)P, sizeof(uint64_t));
  for (uint64_t i = 0; i < n; i++) cnt[hkv[3 * i] >> shift]++;
  uint64_t run = 0;
  for (int p = 0; p < P; p++) {
    first[p] = run;
    run += cnt[p];
    last[p] = run;
  }
  cycle_permute(hkv, 3, P, first, last, ~0ull, shift);
  uint64_t b = 0;
  for (int p = 0; p < P; p++) {
    msd_refine(hkv, 3, b, last[p], 64 - bits, bits);
    b = last[p];
  }
  free(cnt);
  free(first);
  free(last);
}

/* Top level of radix_hash.h:495-587 / radix_sort.h:240-330 as ONE thread (thread_id 0) runs
 * it: per bucket a read cursor rd[] and a write cursor wr[]; an element already in its bucket
 * is compacted to wr; a foreign element is carried along a swap chain until it lands in a
 * bucket that has a blank (rd > wr). */
static void inplace_par_top_t1(uint64_t* a, int W, uint64_t n, int bits, uint64_t* ends) {
  int P = 1 << bits, shift = 64 - bits;
  uint64_t* cnt = (uint64_t*)calloc((size_t)P, sizeof(uint64_t));
  uint64_t* rd = (uint64_t*)malloc((size_t)P * sizeof(uint64_t));
  uint64_t* wr = (uint64_t*)malloc((size_t)P * sizeof(uint64_t));
  uint64_t tmp[3];
  for (uint64_t i = 0; i < n; i++) cnt[a[i * W] >> shift]++;
  uint64_t run = 0;
  for (int p = 0; p < P; p++) {
    rd[p] = wr[p] = run;
    run += cnt[p];
    ends[p] = run;
  }
  int iter = 0;
  while (iter < P) {
    if (rd[iter] >= ends[iter]) {
      iter++;
      continue;
    }
    uint64_t i = rd[iter]++;
    int d = (int)(a[i * W] >> shift);
    if (d == iter) {
      uint64_t j = wr[iter]++;
      if (i != j) el_swap(a + i * W, a + j * W, W);
      continue;
    }
    el_copy(tmp, a + i * W, W);
    for (;;) {
      d = (int)(tmp[0] >> shift);
      if (rd[d] > wr[d]) {
        el_copy(a + (wr[d]++) * W, tmp, W);
        break;
      }
      uint64_t j = rd[d];
      rd[d]++;
      wr[d]++;
      el_swap(tmp, a + j * W, W);
    }
  }
  free(cnt);
  free(rd);
  free(wr);
}

void orc_radix_inplace_par_t1(uint64_t* hkv, uint64_t n, int bits) {
  if (bits < 0) bits = orc_optimal_partition(n);
  int P = 1 << bits;
  uint64_t* ends = (uint64_t*)malloc((size_t)P * sizeof(uint64_t));
  inplace_par_top_t1(hkv, 3, n, bits, ends);
  uint64_t b = 0;
  for (int p = 0; p < P; p++) {
    msd_refine(hkv, 3, b, ends[p], 64 - bits, bits);
    b = ends[p];
  }
  free(ends);
}

void orc_radix_int_inplace_t1(uint64_t* aos, uint64_t n, int bits) {
  if (bits < 0) bits = orc_optimal_partition(n);
  int P = 1 << bits;
  uint64_t* ends = (uint64_t*)malloc((size_t)P * sizeof(uint64_t));
  inplace_par_top_t1(aos, 2, n, bits, ends);
  uint64_t b = 0;
  for (int p = 0; p < P; p++) {
    msd_refine(aos, 2, b, ends[p], 64 - bits, bits);
    b = ends[p];
  }
  free(ends);
}

/* ======================================================================================
 * a10 -- the merge iterator of hashjoin.h:70-180, as a cursor pair (i over R, j over S).
 * ==================================================================================== */
/* hashjoin.h:77-102 and :126-152: skip forward to the next position with equal (hash,key);
 * if either side runs out, both cursors go to their ends. */
static void seek_match(const uint64_t* r, uint64_t nr, const uint64_t* s, uint64_t ns,
                       uint64_t* pi, uint64_t* pj) {
  uint64_t i = *pi, j = *pj;
  for (;;) {
    if (i == nr) {
      j = ns;
      break;
    }
    if (j == ns) {
      i = nr;
      break;
    }
    uint64_t rh = r[3 * i], sh = s[3 * j];
    if (rh < sh) {
      i++;
    } else if (sh < rh) {
      j++;
    } else {
      uint64_t rk = r[3 * i + 1], sk = s[3 * j + 1];
      if (rk == sk) break;
      if (rk < sk)
        i++;
      else
        j++;
    }
  }
  *pi = i;
  *pj = j;
}

uint64_t orc_merge_iterate(const uint64_t* r, uint64_t nr, const uint64_t* s, uint64_t ns,
                           uint64_t* triples, uint64_t cap, uint64_t* sum_out) {
  uint64_t i = 0, j = 0, cnt = 0, sum = 0;
  seek_match(r, nr, s, ns, &i, &j); /* begin(), hashjoin.h:183-186 */
  while (!(i == nr && j == ns)) {   /* != end(), hashjoin.h:165-167 */
    if (triples && cnt < cap) {     /* operator*, hashjoin.h:168-173 */
      triples[3 * cnt] = r[3 * i + 1];
      triples[3 * cnt + 1] = r[3 * i + 2];
      triples[3 * cnt + 2] = s[3 * j + 2];
    }
    sum += r[3 * i + 2] + s[3 * j + 2];
    cnt++;
    /* operator++, hashjoin.h:104-125 */
    if (i + 1 == nr || j + 1 == ns) { /* tail cut :105-114 */
      i = nr;
      j = ns;
      continue;
    }
    if (r[3 * i] == r[3 * (i + 1)]) /* next R has the same hash :115-118 */
      i++;
    else if (s[3 * j] == s[3 * (j + 1)]) /* next S has the same hash :119-122 */
      j++;
    else {
      i++;
      j++;
    }
    seek_match(r, nr, s, ns, &i, &j);
  }
  if (sum_out) *sum_out = sum;
  return cnt;
}

uint64_t orc_hashmergejoin(const uint64_t* r_aos, uint64_t nr, const uint64_t* s_aos,
                           uint64_t ns, int threads, uint64_t* triples, uint64_t cap,
                           uint64_t* sum) {
  uint64_t* rs = (uint64_t*)malloc((size_t)(nr ? nr : 1) * 24);
  uint64_t* ss = (uint64_t*)malloc((size_t)(ns ? ns : 1) * 24);
  orc_radix_non_inplace_par(r_aos, nr, threads, -1, rs); /* hashjoin.h:65 */
  orc_radix_non_inplace_par(s_aos, ns, threads, -1, ss); /* hashjoin.h:67 */
  uint64_t c = orc_merge_iterate(rs, nr, ss, ns, triples, cap, sum);
  free(rs);
  free(ss);
  return c;
}

uint64_t orc_hashmergejoin2(uint64_t* r_hkv, uint64_t nr, uint64_t* s_hkv, uint64_t ns,
                            uint64_t* triples, uint64_t cap, uint64_t* sum) {
  orc_radix_inplace_par_t1(r_hkv, nr, -1); /* hashjoin.h:234 */
  orc_radix_inplace_par_t1(s_hkv, ns, -1); /* hashjoin.h:235 */
  return orc_merge_iterate(r_hkv, nr, s_hkv, ns, triples, cap, sum);
}

/* ======================================================================================
 * a7/a8/a9 -- single-level partition, bucket-local first-wins table, serial probe.
 * ==================================================================================== */
void orc_partition_sizes(const uint64_t* aos, uint64_t n, int bits, uint64_t* sizes) {
  int P = 1 << bits, shift = 64 - bits;
  memset(sizes, 0, (size_t)P * sizeof(uint64_t));
  for (uint64_t i = 0; i < n; i++) sizes[aos[2 * i] >> shift]++; /* partitioned_hash.h:57-60 */
}

/* bucket-local open-addressing table, first insert wins (partitioned_hash.h:166-170) */
typedef struct {
  uint64_t* key;
  uint64_t* val;
  uint8_t* used;
  uint64_t cap; /* power of two */
  uint64_t size;
} ptable;

static void ptable_init(ptable* t, uint64_t expect) {
  uint64_t cap = 8;
  while (cap < 2 * expect + 2) cap <<= 1;
  t->key = (uint64_t*)malloc((size_t)cap * 8);
  t->val = (uint64_t*)malloc((size_t)cap * 8);
  t->used = (uint8_t*)calloc((size_t)cap, 1);
  t->cap = cap;
  t->size = 0;
}
static void ptable_grow(ptable* t);
/* returns slot of key, inserting (key,val) if absent; *was_present tells which */
static uint64_t ptable_upsert(ptable* t, uint64_t key, uint64_t val, int* was_present) {
  if (2 * (t->size + 1) > t->cap) ptable_grow(t);
  uint64_t m = t->cap - 1, s = orc_mix64(key) & m;
  while (t->used[s]) {
    if (t->key[s] == key) {
      *was_present = 1;
      return s;
    }
    s = (s + 1) & m;
  }
  t->used[s] = 1;
  t->key[s] = key;
  t->val[s] = val;
  t->size++;
  *was_present = 0;
  return s;
}
static void ptable_grow(ptable* t) {
  ptable old = *t;
  ptable_init(t, old.cap);
  for (uint64_t s = 0; s < old.cap; s++)
    if (old.used[s]) {
      int p;
      ptable_upsert(t, old.key[s], old.val[s], &p);
    }
  free(old.key);
  free(old.val);
  free(old.used);
}
static void ptable_free(ptable* t) {
  free(t->key);
  free(t->val);
  free(t->used);
}

void orc_partitioned_table_sizes(const uint64_t* aos, uint64_t n, int bits, uint64_t* sizes) {
  int P = 1 << bits, shift = 64 - bits;
  uint64_t* cnt = (uint64_t*)malloc((size_t)P * 8);
  orc_partition_sizes(aos, n, bits, cnt);
  ptable* tabs = (ptable*)malloc((size_t)P * sizeof(ptable));
  for (int p = 0; p < P; p++) ptable_init(&tabs[p], cnt[p]); /* reserve(), :153-160 */
  for (uint64_t i = 0; i < n; i++) {
    int present;
    ptable_upsert(&tabs[aos[2 * i] >> shift], aos[2 * i], aos[2 * i + 1], &present);
  }
  for (int p = 0; p < P; p++) {
    sizes[p] = tabs[p].size;
    ptable_free(&tabs[p]);
  }
  free(tabs);
  free(cnt);
}

uint64_t orc_partitioned_join_sum(const uint64_t* probe, uint64_t n_probe, const uint64_t* build,
                                  uint64_t n_build, int bits, uint64_t* n_found) {
  int P = 1 << bits, shift = 64 - bits;
  /* partition_only(probe side), threads==1: buckets keep input order (:75-79) */
  uint64_t* part = (uint64_t*)malloc((size_t)(n_probe ? n_probe : 1) * 16);
  uint64_t* off = (uint64_t*)malloc((size_t)(P + 1) * 8);
  orc_stable_partition(probe, n_probe, 1, shift, bits, part, off);
  /* partitioned_hash_table(build side) */
  uint64_t* cnt = (uint64_t*)malloc((size_t)P * 8);
  orc_partition_sizes(build, n_build, bits, cnt);
  ptable* tabs = (ptable*)malloc((size_t)P * sizeof(ptable));
  for (int p = 0; p < P; p++) ptable_init(&tabs[p], cnt[p]);
  for (uint64_t i = 0; i < n_build; i++) {
    int present;
    ptable_upsert(&tabs[build[2 * i] >> shift], build[2 * i], build[2 * i + 1], &present);
  }
  /* hashjoin_bench.cc:92-96: sum += r.second + s_tables[i][r.first]; a miss inserts 0 */
  uint64_t sum = 0, found = 0;
  for (int p = 0; p < P; p++) {
    uint64_t before = tabs[p].size;
    (void)before;
    for (uint64_t i = off[p]; i < off[p + 1]; i++) {
      int present;
      uint64_t s = ptable_upsert(&tabs[p], part[2 * i], 0, &present);
      sum += part[2 * i + 1] + tabs[p].val[s];
      found += (uint64_t)present;
    }
  }
  for (int p = 0; p < P; p++) ptable_free(&tabs[p]);
  free(tabs);
  free(cnt);
  free(part);
  free(off);
  if (n_found) *n_found = found;
  return sum;
}

/* ======================================================================================
 * Relational equi-join used to check the GPU executor on inputs with duplicate keys.
 * ==================================================================================== */
typedef struct {
  uint64_t key, val, idx;
} kvi;
static int cmp_kvi(const void* a, const void* b) {
  const kvi *x = (const kvi*)a, *y = (const kvi*)b;
  if (x->key != y->key) return x->key < y->key ? -1 : 1;
  if (x->val != y->val) return x->val < y->val ? -1 : 1;
  return x->idx < y->idx ? -1 : (x->idx > y->idx);
}
static int cmp_triple(const void* a, const void* b) {
  const uint64_t *x = (const uint64_t*)a, *y = (const uint64_t*)b;
  for (int k = 0; k < 3; k++)
    if (x[k] != y[k]) return x[k] < y[k] ? -1 : 1;
  return 0;
}

uint64_t orc_equijoin(const uint64_t* r_aos, uint64_t nr, const uint64_t* s_aos, uint64_t ns,
                      int first_wins, uint64_t* triples, uint64_t cap, orc_checks* checks) {
  kvi* r = (kvi*)malloc((size_t)(nr ? nr : 1) * sizeof(kvi));
  kvi* s = (kvi*)malloc((size_t)(ns ? ns : 1) * sizeof(kvi));
  for (uint64_t i = 0; i < nr; i++) r[i] = (kvi){r_aos[2 * i], r_aos[2 * i + 1], i};
  for (uint64_t i = 0; i < ns; i++) s[i] = (kvi){s_aos[2 * i], s_aos[2 * i + 1], i};
  qsort(r, nr, sizeof(kvi), cmp_kvi);
  qsort(s, ns, sizeof(kvi), cmp_kvi);
  orc_checks c = {0, 0, 0, 0, 0};
  uint64_t i = 0, j = 0, w = 0;
  while (i < nr && j < ns) {
    if (r[i].key < s[j].key) {
      i++;
    } else if (s[j].key < r[i].key) {
      j++;
    } else {
      uint64_t k = r[i].key, ie = i, je = j;
      while (ie < nr && r[ie].key == k) ie++;
      while (je < ns && s[je].key == k) je++;
      uint64_t ib = i, ilim = ie;
      if (first_wins) { /* the build tuple of this key that came first in input order */
        uint64_t best = i;
        for (uint64_t a = i; a < ie; a++)
          if (r[a].idx < r[best].idx) best = a;
        ib = best;
        ilim = best + 1;
      }
      uint64_t group_start = w;
      for (uint64_t a = ib; a < ilim; a++)
        for (uint64_t b = j; b < je; b++) {
          uint64_t m = orc_tmix(k, r[a].val, s[b].val);
          c.n_matches++;
          c.sum_r += r[a].val;
          c.sum_s += s[b].val;
          c.xor_fold ^= m;
          c.mix_sum += m;
          if (triples && w < cap) {
            triples[3 * w] = k;
            triples[3 * w + 1] = r[a].val;
            triples[3 * w + 2] = s[b].val;
          }
          w++;
        }
      (void)group_start;
      i = ie;
      j = je;
    }
  }
  /* groups are emitted r-major/s-minor with both sides sorted by (val), i.e. already in
   * (key, rval, sval) order; a final sort keeps that true under truncation too */
  if (triples) qsort(triples, (size_t)(w < cap ? w : cap), 24, cmp_triple);
  if (checks) *checks = c;
  free(r);
  free(s);
  return c.n_matches;
}
