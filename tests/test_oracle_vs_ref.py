"""Check the C restatement against the REAL reference (oracle/_ref/libhmj_ref.so, compiled from
/root/reference by oracle/Makefile) on seeded random inputs, including duplicate keys and every
(threads, bits) combination the reference's own tests use.  CPU only; skipped if _ref is absent."""
import numpy as np
import pytest


def rel(rng, n, dom=None, lowbits=64):
    if dom is not None:
        k = rng.integers(0, dom, size=n, dtype=np.uint64) * np.uint64(0x9E3779B97F4A7C15)
    elif lowbits < 64:
        k = rng.integers(0, 1 << lowbits, size=n, dtype=np.uint64)
    else:
        k = rng.integers(0, 1 << 63, size=n, dtype=np.uint64) * np.uint64(2) + rng.integers(0, 2, size=n, dtype=np.uint64)
    v = rng.integers(0, 1 << 63, size=n, dtype=np.uint64)
    return np.stack([k, v], 1)


@pytest.mark.parametrize("n", [0, 1, 2, 5, 100, 1000, 12345, 1 << 16])
def test_sorts_match_reference(oracle, reference, n):
    rng = np.random.default_rng(n + 1)
    for kind in ["wide", "low20", "low8", "dups"]:
        a = rel(rng, n, dom=max(1, n // 3)) if kind == "dups" else rel(rng, n, lowbits={"wide": 64, "low20": 20, "low8": 8}[kind])
        for bits in [-1, 1, 2, 3, 7, 14]:
            for T in [1, 3, 8]:
                assert np.array_equal(oracle.radix_non_inplace_par(a, T, bits), reference.radix_non_inplace_par(a, T, bits))
                assert np.array_equal(oracle.radix_int_non_inplace(a, T, bits), reference.radix_int_non_inplace(a, T, bits))
            assert np.array_equal(oracle.radix_int_inplace_t1(a, bits), reference.radix_int_inplace(a, 1, bits))
            h = np.stack([a[:, 0], a[:, 0], a[:, 1]], 1)
            assert np.array_equal(oracle.radix_inplace_seq(h, bits), reference.radix_inplace_seq(h, bits))
            assert np.array_equal(oracle.radix_inplace_par_t1(h, bits), reference.radix_inplace_par(h, 1, bits))
        # multi-threaded in-place variants: tie order may differ, key column may not
        if n:
            assert np.array_equal(reference.radix_int_inplace(a, 4, -1)[:, 0], np.sort(a[:, 0]))


@pytest.mark.parametrize("nr,ns,dom", [(0, 0, 10), (0, 5, 10), (5, 0, 10), (1, 1, 1), (100, 100, 50), (1000, 1000, 300),
                                       (1000, 3000, 10 ** 9), (5000, 5000, 2000), (20000, 20000, 15000), (1 << 16, 1 << 16, 1 << 15)])
def test_joins_match_reference(oracle, reference, nr, ns, dom):
    rng = np.random.default_rng(nr * 7 + ns)
    R, S = rel(rng, nr, dom), rel(rng, ns, dom)
    for T in [1, 4]:
        a, b = oracle.hashmergejoin(R, S, T), reference.hashmergejoin(R, S, T)
        assert a[0] == b[0] and a[1] == b[1] and np.array_equal(a[2], b[2])
    rh = np.stack([R[:, 0], R[:, 0], R[:, 1]], 1)
    sh = np.stack([S[:, 0], S[:, 0], S[:, 1]], 1)
    a, b = oracle.hashmergejoin2(rh, sh), reference.hashmergejoin2(rh, sh, 1)
    assert a[0] == b[0] and a[1] == b[1] and np.array_equal(a[2], b[2])
    # a poor pre-computed hash (many different keys share one): keys unique per relation, the domain of
    # tests/cpp/test_dropin.cc's HashMergeJoin2 cases
    Ru, Su = R[np.unique(R[:, 0], return_index=True)[1]], S[np.unique(S[:, 0], return_index=True)[1]]
    for m in [7, 997]:
        rh = np.stack([Ru[:, 0] % np.uint64(m), Ru[:, 0], Ru[:, 1]], 1)
        sh = np.stack([Su[:, 0] % np.uint64(m), Su[:, 0], Su[:, 1]], 1)
        a, b = oracle.hashmergejoin2(rh, sh), reference.hashmergejoin2(rh, sh, 1)
        assert a[0] == b[0] and a[1] == b[1] and np.array_equal(a[2], b[2])
    for bits in [1, 4, 10]:
        assert oracle.partitioned_join_sum(S, R, bits) == reference.partitioned_join_sum(S, R, 1, bits)
        assert np.array_equal(oracle.partition_sizes(R, bits), reference.partition_only(R, 3, bits))
        assert np.array_equal(oracle.partitioned_table_sizes(R, bits), reference.partitioned_table_sizes(R, 1, bits))


def test_unique_keys_all_formulations_agree(oracle, reference):
    # strgen's invariant (strgen_test.cc:24-33): keys unique per relation -> the sort-merge iterator,
    # the partitioned build+probe and the relational join all describe the same result.
    B = oracle.gen_build(5000)
    P = oracle.gen_probe(4000, 5000, miss_mod=3)
    assert len(np.unique(B[:, 0])) == 5000 and len(np.unique(P[:, 0])) == 4000
    n, sm, t = reference.hashmergejoin(B, P, 3)
    ck, t2 = oracle.equijoin(B, P)
    assert ck["n_matches"] == n and (ck["sum_r"] + ck["sum_s"]) % 2 ** 64 == sm and np.array_equal(t, t2)
    psum, found = reference.partitioned_join_sum(P, B, 4, 10)
    assert found == n and psum == (int(P[:, 1].sum(dtype=np.uint64)) + ck["sum_r"]) % 2 ** 64
    # partition_only with one thread keeps input order inside a bucket == stable pass 1
    sz, cont = reference.partition_only(B, 1, 4, content=True)
    out, off = oracle.stable_partition(B, 60, 4)
    assert np.array_equal(cont, out) and np.array_equal(np.diff(off), sz)


def test_ordered_output_independent_of_threads(reference, oracle):
    # SURVEY 3.3 determinism: ordered sequence identical for num_threads in {1,3,8}
    B = oracle.gen_build(1 << 16)
    P = oracle.gen_probe(1 << 16, 1 << 16)
    outs = [reference.hashmergejoin(B, P, T) for T in (1, 3, 8)]
    assert all(np.array_equal(outs[0][2], x[2]) for x in outs[1:])
    assert np.all(np.diff(outs[0][2][:, 0].astype(np.float64)) > 0)  # ascending key
