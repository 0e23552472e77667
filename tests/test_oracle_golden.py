"""Pin the CPU oracle (oracle/hmj_oracle.c) to the committed golden vectors, which were produced
by the compiled reference (tests/golden/make_golden.py).  CPU only."""
import json
import os

import numpy as np
import pytest

M64 = (1 << 64) - 1


def fnv_rows(a):
    h = 0xCBF29CE484222325
    for b in np.ascontiguousarray(a, np.uint64).tobytes():
        h = ((h ^ b) * 0x100000001B3) & M64
    return h


@pytest.fixture(scope="module")
def G(golden_dir):
    with open(os.path.join(golden_dir, "golden.json")) as f:
        return json.load(f)["cases"]


def test_optimal_partition(oracle, G):
    # radix_hash.h:38-57 known answers (10 @1M, 12 @2^24, 13 @2^26, 14 @2^28)
    for n, k in G["optimal_partition"]:
        assert oracle.optimal_partition(n) == k, n


def test_iterator_edge_cases(oracle, G):
    # hashjoin.h:104-154 staircase / tail-cut semantics
    for c in G["iterator_edge"]:
        R = np.array(c["R"], np.uint64).reshape(-1, 2)
        S = np.array(c["S"], np.uint64).reshape(-1, 2)
        n, sm, t = oracle.hashmergejoin(R, S, 1)
        assert n == c["n"] and sm == c["sum"]
        assert t.tolist() == c["triples"]


def test_radix_hash_descending_shapes(oracle, G):
    # radix_hash_test.cc:40-120 shapes
    for c in G["radix_hash_desc"]:
        n, start = c["n"], c["start"]
        keys = np.arange(start + n - 1, start - 1, -1, dtype=np.uint64)
        a = np.stack([keys, keys], 1)
        out = oracle.radix_non_inplace_par(a, c["threads"], c["bits"])
        assert np.array_equal(out[:, 0], np.arange(start, start + n, dtype=np.uint64))
        assert oracle.fnv1a_triples(out) == c["fnv_hkv"]


def test_radix_inplace_par_simple(oracle, G):
    # radix_hash_test.cc:227-242
    keys = np.array([(i | (1 << 63)) if i % 2 else i for i in range(1024)], np.uint64)
    hkv = np.stack([keys, keys, keys], 1)
    out = oracle.radix_inplace_par_t1(hkv, 1)
    c = G["radix_inplace_par_simple"]
    assert oracle.fnv1a_triples(out) == c["fnv_hkv"]
    assert out[:4, 0].tolist() == c["first"] and out[-2:, 0].tolist() == c["last"]


def test_radix_int_random(oracle, G, golden_dir):
    # radix_sort_test.cc:27-68 shape
    c = G["radix_int_random"]
    a = np.load(os.path.join(golden_dir, c["input"]))
    assert fnv_rows(a) == c["fnv_input"]
    srt = np.sort(a[:, 0])
    x = oracle.radix_int_non_inplace(a, 8, -1)
    assert np.array_equal(x[:, 0], srt) and fnv_rows(x) == c["non_inplace_T8"]
    x = oracle.radix_int_non_inplace(a, 1, 10)
    assert fnv_rows(x) == c["non_inplace_T1_bits10"]
    x = oracle.radix_int_inplace_t1(a, -1)
    assert np.array_equal(x[:, 0], srt) and fnv_rows(x) == c["inplace_T1"]
    x = oracle.radix_non_inplace_par(a, 3, -1)
    assert oracle.fnv1a_triples(x) == c["hash_non_inplace_T3"]


def test_partition_15(oracle, G):
    # partitioned_hash_test.cc:14-46: sizes {5,10}
    c = G["partition_15"]
    src = np.array(c["src"], np.uint64)
    assert oracle.partition_sizes(src, 1).tolist() == c["partition_only_T2_bits1"] == [5, 10]
    assert oracle.partitioned_table_sizes(src, 1).tolist() == c["partition_table_T1_bits1"] == [5, 10]


def test_generated_joins(oracle, G, golden_dir):
    for c in G["gen_join"]:
        nb, npb, miss = c["n_build"], c["n_probe"], c["miss_mod"]
        if nb > (1 << 20):
            continue
        B = oracle.gen_build(nb)
        P = oracle.gen_probe(npb, nb, miss_mod=miss)
        if c["fnv_build"] is not None:
            assert fnv_rows(B) == c["fnv_build"] and fnv_rows(P) == c["fnv_probe"]
        n, sm, t = oracle.hashmergejoin(B, P, 4)
        assert n == c["n"] and sm == c["sum"]
        assert oracle.checks_of_triples(t) == c["checks"]
        assert oracle.fnv1a_triples(t) == c["fnv_ordered"]
        if "triples" in c:
            assert np.array_equal(t, np.load(os.path.join(golden_dir, c["triples"])))
        # unique keys on both sides: relational join == reference iterator output (P2)
        ck, t2 = oracle.equijoin(B, P)
        assert ck == c["checks"] and np.array_equal(t2, t)
        assert list(oracle.partitioned_join_sum(P, B, 10)) == c["psum_T1_bits10"]


def test_duplicate_build_keys(oracle, G, golden_dir):
    for c in G["dup_partitioned"]:
        z = np.load(os.path.join(golden_dir, c["file"]))
        B, P = z["build"], z["probe"]
        s, f = oracle.partitioned_join_sum(P, B, 10)
        assert (s, f) == (c["psum"], c["pfound"])
        n, sm, t = oracle.hashmergejoin(B, P, 1)
        assert (n, sm, oracle.fnv1a_triples(t)) == (c["hmj_n"], c["hmj_sum"], c["hmj_fnv"])
        # first-wins relational join reproduces the partitioned bench sum:
        #   sum = sum(all probe vals) + sum(matched first build vals)
        ck, _ = oracle.equijoin(B, P, first_wins=True)
        assert (int(P[:, 1].sum(dtype=np.uint64)) + ck["sum_r"]) & M64 == c["psum"]


def test_mix64_bijection(oracle):
    for x in [0, 1, 12345, M64, 0x243F6A8885A308D3, 1 << 63]:
        assert oracle.unmix64(oracle.mix64(x)) == x
        assert oracle.mix64(oracle.unmix64(x)) == x


def test_strgen_restatement_matches_the_goldens(golden_dir):
    # create_strvec (strgen.cc:27-61) restated twice -- C++ (oracle/strgen_restated.h, which fed the compiled
    # reference when the goldens were made) and Python (oracle/pyoracle.py) -- over the word-list fixture: the
    # Python one must regenerate the very relations the goldens were computed from, and (strgen_test.cc:24-33)
    # all keys of a relation are distinct
    import json

    from oracle.pyoracle import create_strvec, fnv_relation

    words = open(os.path.join(golden_dir, "words.txt")).read().split("\n")[:-1]
    assert len(words) == 1200 and len(set(words)) == 1200 and not any("-" in w for w in words)
    cases = json.load(open(os.path.join(golden_dir, "golden.json")))["cases"]["strgen_join"]
    assert [c["n"] for c in cases] == [2, 1000, 1 << 12, 1 << 16, 1 << 18, 10 ** 6]
    for c in cases:
        assert c["count"] == c["n"] == c["distinct"]  # every key matches exactly once (same key set, two orders)
        if c["n"] > 1 << 16:
            continue
        r, s = create_strvec(c["n"], words, c["seed_r"]), create_strvec(c["n"], words, c["seed_s"])
        assert fnv_relation(r) == c["fnv_r"] and fnv_relation(s) == c["fnv_s"], c["n"]
        assert len({k for k, _ in r}) == c["n"] and {k for k, _ in r} == {k for k, _ in s}
        # the join the reference computed: each key pairs r's payload with s's payload -> the sum is 2 x sum of payloads
        assert (2 * sum(v for _, v in r)) & ((1 << 64) - 1) == c["sum"]
    # strgen_test.cc's own size
    big = create_strvec(1 << 18, words, 1)
    assert len({k for k, _ in big}) == 1 << 18
