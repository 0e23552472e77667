/*
 * TEST INFRASTRUCTURE ONLY -- CPU oracle for the radix-partitioned join hot path.
 *
 * A from-scratch plain-C restatement of the reference algorithms on the path named by
 * BASELINE.json.  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load this library; the product (hashmergejoin_amd/) never does.
 *
 * PARITY PINNED: every function below is checked against the real reference compiled from
 * /root/reference (oracle/_ref/libhmj_ref.so, see oracle/ref_driver.cc) in
 * tests/test_oracle_vs_ref.py, and against committed golden vectors generated from that
 * same compiled reference (tests/golden/, generator script tests/golden/make_golden.py).
 *
 * Layouts: "aos"  = n x { uint64 key; uint64 val }            (std::pair<u64,u64>, 16 B)
 *          "hkv"  = n x { uint64 hash; uint64 key; uint64 val } (logical order of the
 *                    reference's std::tuple<size_t,Key,Value>; 24 B)
 *          "triples" = n x { uint64 key; uint64 rval; uint64 sval }
 * Hash is std::hash<uint64_t> == identity.
 */
#ifndef HMJ_ORACLE_H
#define HMJ_ORACLE_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- synthetic relations ---------------------------------------- */
#define ORC_SEED_B 0x243F6A8885A308D3ull
#define ORC_PI_A 0x9E3779B1ull
#define ORC_PI_B 12345ull
#define ORC_VAL_XOR 0x9E3779B97F4A7C15ull

uint64_t orc_mix64(uint64_t x);   /* bijective splitmix64 finalizer */
uint64_t orc_unmix64(uint64_t x); /* its inverse */
/* build side: key = mix64(i + seed), val = i, i in [start, start+n) */
void orc_gen_build(uint64_t* aos, uint64_t n, uint64_t start, uint64_t seed);
/* probe side: j in [start, start+n): idx = (A*j + B) mod n_build (+ n_build if miss_mod>0 and
 * j % miss_mod == 0, i.e. a key outside the build set); key = mix64(idx + seed),
 * val = j ^ ORC_VAL_XOR */
void orc_gen_probe(uint64_t* aos, uint64_t n, uint64_t start, uint64_t n_build, uint64_t seed,
                   uint64_t miss_mod);
/* skewed build side: rank = lower_bound(thr, mix64(i ^ zseed)) over a caller-made table of
 * `domain` ascending u64 thresholds (CDF * 2^64, e.g. Zipf theta=0.9); key = mix64(rank + seed),
 * val = i.  Integer-only, so the GPU generator reproduces it bit for bit. */
void orc_gen_from_cdf(uint64_t* aos, uint64_t n, uint64_t start, const uint64_t* thr,
                      uint64_t domain, uint64_t seed, uint64_t zseed);
/* uniform over `domain` values: key = mix64((mix64(j ^ zseed) % domain) + seed), val = j^XOR */
void orc_gen_uniform_domain(uint64_t* aos, uint64_t n, uint64_t start, uint64_t domain,
                            uint64_t seed, uint64_t zseed);

/* ---- result checksums (parity definition P1) --------------------- */
typedef struct {
  uint64_t n_matches;
  uint64_t sum_r;    /* sum of rval over matches, mod 2^64 */
  uint64_t sum_s;    /* sum of sval over matches (sum_r + sum_s == hashjoin_bench.cc:132) */
  uint64_t xor_fold; /* XOR of tmix(key,rval,sval) */
  uint64_t mix_sum;  /* sum of tmix(key,rval,sval) mod 2^64 */
} orc_checks;
uint64_t orc_tmix(uint64_t key, uint64_t rval, uint64_t sval);
void orc_checks_of_triples(const uint64_t* triples, uint64_t n, orc_checks* out);
/* order-sensitive FNV-1a (64-bit) over the triples' bytes, little endian */
uint64_t orc_fnv1a_triples(const uint64_t* triples, uint64_t n);

/* ---- a2: radix_hash.h:38-57 ------------------------------------------------------------ */
int orc_optimal_partition(uint64_t input_num);

/* ---- a3+a4+a5: radix_hash.h:294-423 (bits < 0 -> optimal_partition) -------------------- */
void orc_radix_non_inplace_par(const uint64_t* aos, uint64_t n, int threads, int bits,
                               uint64_t* out_hkv);
/* pass 1 alone (radix_hash.h:313-345 / radix_sort.h:418-449): stable scatter on
 * digit = (key >> shift) & (2^bits - 1); offsets[2^bits + 1] */
void orc_stable_partition(const uint64_t* aos, uint64_t n, int threads, int shift, int bits,
                          uint64_t* out_aos, uint64_t* offsets);
/* radix_hash.h:425-493 and :495-654 (threads == 1 schedule); in place on hkv */
void orc_radix_inplace_seq(uint64_t* hkv, uint64_t n, int bits);
void orc_radix_inplace_par_t1(uint64_t* hkv, uint64_t n, int bits);

/* ---- a6: radix_sort.h:400-522 and :240-398 (threads == 1 schedule) --------------------- */
void orc_radix_int_non_inplace(const uint64_t* aos, uint64_t n, int threads, int bits,
                               uint64_t* out_aos);
void orc_radix_int_inplace_t1(uint64_t* aos, uint64_t n, int bits);

/* ---- a10: hashjoin.h:56-68 + merge iterator :70-180 ------------------------------------ */
/* merge two hkv arrays already sorted by (hash,key); returns tuples yielded; writes at most
 * cap triples; *sum = sum(rval + sval) */
uint64_t orc_merge_iterate(const uint64_t* r_hkv, uint64_t nr, const uint64_t* s_hkv,
                           uint64_t ns, uint64_t* triples, uint64_t cap, uint64_t* sum);
uint64_t orc_hashmergejoin(const uint64_t* r_aos, uint64_t nr, const uint64_t* s_aos,
                           uint64_t ns, int threads, uint64_t* triples, uint64_t cap,
                           uint64_t* sum);
/* a11: hashjoin.h:201-363 on pre-hashed tuples (sorted in place, threads == 1 schedule) */
uint64_t orc_hashmergejoin2(uint64_t* r_hkv, uint64_t nr, uint64_t* s_hkv, uint64_t ns,
                            uint64_t* triples, uint64_t cap, uint64_t* sum);

/* ---- a7+a8+a9: partitioned_hash.h:36-215 + hashjoin_bench.cc:92-96 (threads == 1) ------ */
void orc_partition_sizes(const uint64_t* aos, uint64_t n, int bits, uint64_t* sizes);
void orc_partitioned_table_sizes(const uint64_t* aos, uint64_t n, int bits, uint64_t* sizes);
uint64_t orc_partitioned_join_sum(const uint64_t* probe_aos, uint64_t n_probe,
                                  const uint64_t* build_aos, uint64_t n_build, int bits,
                                  uint64_t* n_found);

/* ---- relational equi-join (what the GPU executor implements for duplicate keys) -------- */
/* R = build, S = probe.  first_wins == 0: cross product per key.  first_wins != 0: each
 * probe tuple pairs with the earliest (input order) build tuple of its key
 * (partitioned_hash.h:166-170 insert semantics).  Triples are produced sorted by
 * (key, rval, sval); at most cap are written; checks cover all of them. */
uint64_t orc_equijoin(const uint64_t* r_aos, uint64_t nr, const uint64_t* s_aos, uint64_t ns,
                      int first_wins, uint64_t* triples, uint64_t cap, orc_checks* checks);

#ifdef __cplusplus
}
#endif
#endif
