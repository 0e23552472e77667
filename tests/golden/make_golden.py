#!/usr/bin/env python3
"""Generate tests/golden/* from the REAL reference (oracle/_ref/libhmj_ref.so).

Run in the build container, where /root/reference exists and `make -C oracle` has compiled it:
    python tests/golden/make_golden.py
Inputs come from the seeded synthetic generators; every fixture stores the
generator parameters, an FNV-1a checksum of the generated input (so generator drift is caught) and
the reference's outputs.  Fixtures are data only -- no reference source text is stored.
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle.pyoracle import Oracle, Reference  # noqa: E402

o, ref = Oracle(), Reference()
assert ref.available, "build oracle/_ref first: make -C oracle ref"
M64 = (1 << 64) - 1


def fnv_rows(a):
    """FNV-1a over the little-endian bytes of a uint64 array (any width)."""
    h = 0xCBF29CE484222325
    for b in np.ascontiguousarray(a, np.uint64).tobytes():
        h = ((h ^ b) * 0x100000001B3) & M64
    return h


def fnv_fast(a):
    # same function as fnv_rows, via the oracle's C loop (arrays reshaped to triples-compatible)
    a = np.ascontiguousarray(a, np.uint64).reshape(-1)
    pad = (-len(a)) % 3
    assert pad == 0, "use widths that are multiples of 3 words or fnv_rows"
    return o.fnv1a_triples(a.reshape(-1, 3))


G = {"_about": "golden vectors from the compiled reference; see make_golden.py", "cases": {}}
C = G["cases"]

# (1) SURVEY 3.3 iterator edge cases: exact sequences ------------------------------------------
edge = [
    ([(5, 1), (5, 2), (9, 3)], [(5, 10), (5, 20), (9, 30)]),
    ([(5, 1)], [(5, 10), (5, 20)]),
    ([(5, 1), (7, 2)], [(5, 10), (5, 20), (7, 30)]),
    ([(5, 1), (5, 2), (7, 3)], [(5, 10), (7, 30)]),
    ([(1, 1)], [(2, 2)]),
    ([], [(2, 2)]),
    ([(3, 1), (3, 2), (3, 3)], [(3, 7), (3, 8), (3, 9), (4, 1)]),
]
C["iterator_edge"] = []
for R, S in edge:
    n, sm, t = ref.hashmergejoin(np.array(R, np.uint64).reshape(-1, 2), np.array(S, np.uint64).reshape(-1, 2), 1)
    C["iterator_edge"].append({"R": R, "S": S, "n": n, "sum": sm, "triples": t.tolist()})

# (2) radix_hash_test.cc shapes: descending keys, identity hash ---------------------------------
C["radix_hash_desc"] = []
for n, bits, T in [(5, 3, 1), (5, 1, 1), (5, 2, 1), (12345, 1, 8), (12345, 14, 8), (1 << 18, -1, 8), (1 << 18, -1, 4)]:
    start = 0 if (n == 5 and bits == 3) else 1  # full_sort uses 4..0, the others n..1
    keys = np.arange(start + n - 1, start - 1, -1, dtype=np.uint64)
    a = np.stack([keys, keys], 1)
    out = ref.radix_non_inplace_par(a, T, bits)
    assert np.array_equal(out[:, 0], np.sort(keys))
    C["radix_hash_desc"].append({"n": n, "bits": bits, "threads": T, "start": start, "fnv_hkv": fnv_fast(out)})
# radix_inplace_par_test.simple_input: 1024 keys, half with bit 63 set, T=8, bits=1
keys = np.array([(i | (1 << 63)) if i % 2 else i for i in range(1024)], np.uint64)
hkv = np.stack([keys, keys, keys], 1)
out = ref.radix_inplace_par(hkv, 1, 1)
C["radix_inplace_par_simple"] = {"fnv_hkv": fnv_fast(out), "first": out[:4, 0].tolist(), "last": out[-2:, 0].tolist()}

# (3) radix_sort_test.cc shape: 2^16 random u64 keys (numpy PCG64 seed 20180601), val = i --------
rng = np.random.default_rng(20180601)
keys = rng.integers(0, 1 << 63, size=1 << 16, dtype=np.uint64) * np.uint64(2) + rng.integers(0, 2, size=1 << 16, dtype=np.uint64)
a = np.stack([keys, np.arange(1 << 16, dtype=np.uint64)], 1)
np.save(os.path.join(HERE, "rand64k_input.npy"), a)
C["radix_int_random"] = {
    "input": "rand64k_input.npy",
    "fnv_input": fnv_rows(a),
    "non_inplace_T8": fnv_rows(ref.radix_int_non_inplace(a, 8, -1)),
    "non_inplace_T1_bits10": fnv_rows(ref.radix_int_non_inplace(a, 1, 10)),
    "inplace_T1": fnv_rows(ref.radix_int_inplace(a, 1, -1)),
    "hash_non_inplace_T3": fnv_fast(ref.radix_non_inplace_par(a, 3, -1)),
}

# (4) partitioned_hash_test.cc: 15-key top-bit routing ------------------------------------------
top = 1 << 63
src = []
for i in range(5):
    src += [(i, i), (i | top, i | top), (i | top | 1024, i | top | 1024)]
src = np.array(src, np.uint64)
C["partition_15"] = {
    "src": src.tolist(),
    "partition_only_T2_bits1": ref.partition_only(src, 2, 1).tolist(),
    "partition_table_T1_bits1": ref.partitioned_table_sizes(src, 1, 1).tolist(),
}

# (5) generator-based unique-key joins -----------------------------------------------------------
C["gen_join"] = []
for nb, npb, miss in [(1 << 10, 1 << 10, 0), (1 << 12, 1 << 12, 0), (12345, 12345, 0), (1 << 12, 1 << 12, 2),
                      (1 << 12, 3000, 3), (1 << 16, 1 << 16, 0), (1 << 16, 1 << 16, 2), (1 << 20, 1 << 20, 0),
                      (1 << 20, 1 << 20, 2), (10 ** 6, 10 ** 6, 0)]:
    B = o.gen_build(nb)
    P = o.gen_probe(npb, nb, miss_mod=miss)
    n, sm, t = ref.hashmergejoin(B, P, 8)
    ck = o.checks_of_triples(t)  # checksum arithmetic only; the triples themselves are the reference's
    assert (ck["sum_r"] + ck["sum_s"]) & M64 == sm and ck["n_matches"] == n
    case = {"n_build": nb, "n_probe": npb, "miss_mod": miss, "fnv_build": fnv_rows(B) if nb <= 1 << 16 else None,
            "fnv_probe": fnv_rows(P) if npb <= 1 << 16 else None, "n": n, "sum": sm, "checks": ck,
            "fnv_ordered": o.fnv1a_triples(t),
            "psum_T1_bits10": list(ref.partitioned_join_sum(P, B, 1, 10))}
    if nb <= 12345:
        name = "join_%d_%d_m%d.npy" % (nb, npb, miss)
        np.save(os.path.join(HERE, name), t)
        case["triples"] = name
    C["gen_join"].append(case)

# (6) duplicate build keys, partitioned (first insert wins, miss -> 0) semantics, T=1 -------------
rng = np.random.default_rng(7)
C["dup_partitioned"] = []
for nb, npb, dom in [(200, 300, 50), (5000, 8000, 1200), (1 << 14, 1 << 15, 3000)]:
    bk = rng.integers(0, dom, size=nb, dtype=np.uint64)
    pk = rng.integers(0, dom + dom // 4, size=npb, dtype=np.uint64)
    B = np.stack([np.array([o.mix64(int(k)) for k in bk], np.uint64), np.arange(nb, dtype=np.uint64) + np.uint64(1000)], 1)
    P = np.stack([np.array([o.mix64(int(k)) for k in pk], np.uint64), np.arange(npb, dtype=np.uint64) * np.uint64(3)], 1)
    name = "dup_%d_%d.npz" % (nb, npb)
    np.savez_compressed(os.path.join(HERE, name), build=B, probe=P)
    s, f = ref.partitioned_join_sum(P, B, 1, 10)
    n, sm, t = ref.hashmergejoin(B, P, 1)
    C["dup_partitioned"].append({"file": name, "psum": s, "pfound": f, "hmj_n": n, "hmj_sum": sm,
                                 "hmj_fnv": o.fnv1a_triples(t)})

# (7) std::string keys through the reference (synthetic strgen-shaped relations, ref_driver.cc) -----
C["string_join"] = []
for nr, ns, seed in [(1000, 1000, 1), (5000, 3000, 2), (200000, 150000, 3)]:
    n, sm, pairs = ref.hashmergejoin_str(nr, ns, seed, 4)
    C["string_join"].append({"nr": nr, "ns": ns, "seed": seed, "n": n, "sum": sm, "fnv_pairs": fnv_rows(pairs)})

# (8) the reference's benchmark relations: r = create_strvec(n), s = create_strvec(n) (hashjoin_bench.cc:112-113)
# from the restated generator over the word-list fixture (oracle/strgen_restated.h, tests/golden/words.txt),
# joined by the compiled reference.  10^6 is BASELINE.json configs[0]; 2^18 is strgen_test.cc's size.
WORDS = os.path.join(HERE, "words.txt")
C["strgen_join"] = []
for n in [2, 1000, 1 << 12, 1 << 16, 1 << 18, 10 ** 6]:
    g = ref.hashmergejoin_strgen(WORDS, n, 1, 2, 8)
    assert g["n"] == n and g["distinct"] == n  # same key set on both sides; strgen_test.cc:24-33 uniqueness
    C["strgen_join"].append({"n": n, "seed_r": 1, "seed_s": 2, "count": g["n"], "sum": g["sum"], "fnv_pairs": fnv_rows(g["pairs"]),
                             "fnv_r": g["fnv_r"], "fnv_s": g["fnv_s"], "distinct": g["distinct"]})

# (9) full-size unique-key joins: the reference's own checksums at sizes the GPU tests otherwise check by the
# generator's closed forms only (2^24: ~1 s; 2^26: ~20 s and ~10 GiB in the build container)
C["gen_join_full"] = []
for log2n, miss in [(24, 0), (24, 3), (26, 0)]:
    nn = 1 << log2n
    B = o.gen_build(nn)
    P = o.gen_probe(nn, nn, miss_mod=miss)
    n, sm, t = ref.hashmergejoin(B, P, 8, cap=nn)
    ck = o.checks_of_triples(t)  # checksum arithmetic only; the triples themselves are the reference's
    assert (ck["sum_r"] + ck["sum_s"]) & M64 == sm and ck["n_matches"] == n
    keys = t[:, 0]
    assert bool(np.all(keys[1:] > keys[:-1]))  # iteration order = ascending key
    C["gen_join_full"].append({"log2n": log2n, "miss_mod": miss, "n": n, "sum": sm, "checks": ck})
    del B, P, t, keys

# optimal_partition table --------------------------------------------------------------------------
ns = [0, 1, 5, 63, 64, 65, 1000, 4095, 4096, 12345, 10 ** 6, 1 << 16, 1 << 18, 1 << 20, 1 << 24, 1 << 26, 1 << 28, 1 << 30, 1 << 31,
      10 ** 7, 10 ** 9]
C["optimal_partition"] = [[n, ref.optimal_partition(n)] for n in ns]

with open(os.path.join(HERE, "golden.json"), "w") as f:
    json.dump(G, f, indent=1)
print("wrote golden.json with", {k: (len(v) if isinstance(v, list) else 1) for k, v in C.items()})
