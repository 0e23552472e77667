#!/usr/bin/env python3
"""bench.py -- probe-side tuples/s of the radix-partitioned hash join on MI355X.

A "step" is one whole pass of the hot path over one batch of synthetic relations that are already
resident in HBM: radix-partition R, radix-partition S, bucket-local build + probe in count/sum mode
(the reduction hashjoin_bench.cc:131-133 performs).  value = probe rows of all ranks x steps / time.

N=1 workload: BASELINE.json configs[2]: |R|=|S|=2^28 u64 key / 8 B payload, 100 % match.
N>1: weak scaling, 2^28 rows per relation per GPU (N=8 -> |R|=|S|=2^31, configs[3]); each step is one
hmj_exchange_join_u64_device: the first radix pass on every rank, its digit ranges as owners, RCCL grouped send/recv
rounds of digit ranges, one join per arrived round.  `python bench.py --gpus N` starts its own N ranks when no
launcher did (WORLD_SIZE unset).

Prints ONE JSON line (rank 0).  Extra objects: "roofline" (dominant kernels, HIP-event timed inside the
timed region), "roofline_probe" (the build+probe kernel), "cpu_baseline" (the compiled reference timed on
the host cores, N=1 only), "extra" (N=1: other BASELINE configs / modes timed in the same run).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def usable_cores():
    """(threads to use, cores in the affinity mask).  The smaller of the affinity mask and the cgroup CPU quota."""
    try:
        aff = len(os.sched_getaffinity(0))
    except Exception:
        aff = os.cpu_count() or 1
    use = aff
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            use = max(1, min(use, int(int(q) / int(p))))
    except Exception:
        pass
    if os.environ.get("HMJ_CPU_THREADS"):
        use = max(1, int(os.environ["HMJ_CPU_THREADS"]))
    return use, aff


def cpu_baseline(log2n, threads, affinity):
    """The reference's own pthread CPU path on a bounded sample of the same workload, timed on this box's host
    cores as SURVEY.md 8(d) prescribes: all cores the process may use, one warm-up + best of 5, buffers
    pre-faulted (the relations are generated and copied into the reference's vectors before the clock starts).
    kind "reference": oracle/_ref/libhmj_ref.so is the real reference compiled from its own headers
    (oracle/Makefile).  Falls back to the single-thread C port where that library is absent."""
    from oracle.pyoracle import Oracle, Reference

    orc, ref = Oracle(), Reference()
    n = 1 << log2n
    B, P = orc.gen_build(n), orc.gen_probe(n, n)
    out = {"cores": threads, "affinity_cores": affinity, "unit": "probe tuples/s", "runs": "1 warm-up + best of 5",
           "sample": "|R|=|S|=2^%d u64 key / 8 B payload, same generator, 100%% match" % log2n}

    def best_of(fn, reps=5):
        fn()  # warm-up (also faults the destination buffers in)
        best = None
        for _ in range(reps):
            t0 = time.perf_counter()
            fn()
            dt = time.perf_counter() - t0
            best = dt if best is None else min(best, dt)
        return best

    if ref.available:
        rh, sh, oh = ref.pairs_new(B), ref.pairs_new(P), ref.pairs_new(B)
        res = {}

        def join():
            res["v"] = ref.hashmergejoin_pairs(rh, sh, threads)  # ctor + iterate, hashjoin_bench.cc:126-133

        best = best_of(join)
        assert res["v"][0] == n
        out.update(kind="reference", value=n / best, seconds=best, what="HashMergeJoin ctor + iterate (hashjoin_bench.cc:126-133)")
        # what radix_bench_par times for u64 keys (radix_bench_par.cc:126-127 and :96)
        out["radix_int_non_inplace_keys_per_s"] = n / best_of(lambda: ref.radix_int_non_inplace_pairs(rh, oh, threads))
        m = min(n, 1 << 22)  # the in-place sort is much slower: shorter sample, best of 3 (each run sorts a fresh copy)
        out["radix_int_inplace_keys_per_s"] = m / best_of(lambda: ref.radix_int_inplace(B[:m], threads), reps=3)
        out["radix_int_inplace_sample_log2"] = 22
        # partition_only + partitioned_hash_table + probe (hashjoin_bench.cc:88-96): mutex-guarded node tables, ~3 M rows/s --
        # a 2^22-row sample, one timed repetition (at 2^24 it took 17 of the bench's 32 s)
        mp = min(n, 1 << 22)
        Bp, Pp = orc.gen_build(mp), orc.gen_probe(mp, mp)
        out["partitioned_build_probe_tuples_per_s"] = mp / best_of(lambda: ref.partitioned_join_sum(Pp, Bp, threads, 10), reps=1)
        out["partitioned_build_probe_sample_log2"] = 22
        for h in (rh, sh, oh):
            ref.pairs_free(h)
        # BASELINE configs[0]: the reference's own benchmark relations (two create_strvec(10^6), std::string keys), construct
        # + iterate timed as hashjoin_bench.cc:120-134 does; the GPU drop-in on the same relations: extra.configs0_strgen_1M_ms
        words = os.path.join(ROOT, "tests", "golden", "words.txt")
        try:
            sec, ck = ref.hashmergejoin_strgen_timed(words, 1000000, threads, reps=3)
            out["strgen_1M_seconds"] = sec
            out["strgen_1M_tuples_per_s"] = 1000000 / sec
            out["strgen_1M_checks"] = {"count": ck[0], "sum": ck[1], "fnv_pairs": ck[2]}
            want = [c for c in json.load(open(os.path.join(ROOT, "tests", "golden", "golden.json")))["cases"]["strgen_join"] if c["n"] == 1000000][0]
            out["strgen_1M_equals_golden"] = ck == (want["count"], want["sum"], want["fnv_pairs"])
        except Exception as e:
            out["strgen_1M_seconds"] = None
            out["strgen_1M_error"] = repr(e)
    else:
        best = best_of(lambda: orc.hashmergejoin(B, P, 1, cap=0), reps=2)
        out.update(kind="port", cores=1, value=n / best, seconds=best, what="C restatement of HashMergeJoin ctor + iterate, 1 thread")
    return out


def hipmalloc_cost_ms(sizes_gib=(1, 4)):
    """What this box charges for creating device memory, in the bench process, before the library reserves its workspace:
    hipMalloc + hipFree of 1 and 4 GiB straight through the HIP runtime (no caching allocator).  The first call of a join
    context pays this per GB of partition buffers (18 GB at 2^28 x 2^28 rows): 0.3 ms in all on some boxes, 26 ms per GB on
    the driver's round-4 box -- a property of the box, reported so that `reserve_ms` can be read."""
    import ctypes as C

    try:
        hip = C.CDLL(None)
        if not hasattr(hip, "hipMalloc"):
            hip = C.CDLL("libamdhip64.so")
        hip.hipMalloc.restype = C.c_int
        hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
        hip.hipFree.restype = C.c_int
        hip.hipFree.argtypes = [C.c_void_p]
        out = {}
        for g in sizes_gib:
            p = C.c_void_p()
            t0 = time.perf_counter()
            rc = hip.hipMalloc(C.byref(p), g << 30)
            t1 = time.perf_counter()
            if rc != 0:
                return {"error": "hipMalloc(%d GiB) returned %d" % (g, rc)}
            hip.hipFree(p)
            out["%dGiB" % g] = {"hipMalloc_ms": round((t1 - t0) * 1e3, 3), "hipFree_ms": round((time.perf_counter() - t1) * 1e3, 3)}
        big = out["%dGiB" % sizes_gib[-1]]["hipMalloc_ms"]
        out["ms_per_GB"] = round(big / (sizes_gib[-1] * 1.073741824), 3)
        return out
    except Exception as e:  # reporting only
        return {"error": repr(e)}


def extra_runs(ex, H, torch):
    """Other BASELINE configs and operator modes, timed by the same run (best of 3 after a warm-up, wall clock
    around the blocking call): configs[1] with its stated 10-bit plan and with the planner's, configs[4], and the
    ordered mode the C++ operator uses."""
    import numpy as np

    def timed(fn, reps=3):
        fn()
        best = None
        for _ in range(reps):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            r = fn()
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) * 1e3
            best = dt if best is None else min(best, dt)
        return round(best, 3), r

    out = {}
    roofs = {}  # per entry: the algorithmic minimum of the call (every input row read once, every result row written once),
    # the whole call's wall-clock ms and the fraction of the 8 TB/s peak those give -- whole calls, not kernels

    def roof(key, ms, nb_, np_, rows=0, sort=False):
        nbytes = 32 * nb_ if sort else 16 * (nb_ + np_) + 24 * rows
        roofs[key] = {"algorithmic_bytes": nbytes, "ms": ms, "frac": round(nbytes / (ms * 1e-3) / (HBM_PEAK_GBS * 1e9), 4)}

    # BASELINE configs[0] through the C++ drop-in: r = create_strvec(10^6), s = create_strvec(10^6) (strgen.cc:27-61 restated
    # over the word-list fixture), HashMergeJoin<KeyValVec::iterator, ...> constructed + iterated (hashjoin_bench.cc:120-134):
    # std::hash<std::string> on the host, {hash, row} pairs joined on the GPU, collisions resolved on the host.  A process
    # of its own (tests/cpp/strgen_bench); count / sum / ordered FNV must equal the compiled reference's (the golden).
    sb = os.path.join(ROOT, "tests", "cpp", "strgen_bench")
    if os.path.exists(sb):
        import subprocess

        pr = subprocess.run([sb, os.path.join(ROOT, "tests", "golden", "words.txt"), "1000000", "5"], stdout=subprocess.PIPE,
                            stderr=subprocess.PIPE, timeout=300)
        js = [l for l in pr.stdout.decode().splitlines() if l.startswith("{")]
        assert pr.returncode == 0 and js, ("strgen_bench failed", pr.returncode, pr.stderr.decode()[-500:])
        sg = json.loads(js[-1])
        want = [c for c in json.load(open(os.path.join(ROOT, "tests", "golden", "golden.json")))["cases"]["strgen_join"] if c["n"] == 1000000][0]
        assert (sg["count"], sg["sum"], sg["fnv"]) == (want["count"], want["sum"], want["fnv_pairs"]), ("configs[0] differs from the golden", sg, want)
        assert (sg["fnv_r"], sg["fnv_s"]) == (want["fnv_r"], want["fnv_s"]), "configs[0]: generated relations differ from the golden's"
        out["configs0_strgen_1M_ms"] = sg["ms"]
        out["configs0_strgen_1M"] = {"ms_ctor": sg["ms_ctor"], "ms_iterate": sg["ms_iterate"], "ms_first_call": sg["ms_first_call"],
                                     "host_threads": sg["host_threads"], "tuples_per_s": round(1e6 / (sg["ms"] * 1e-3)),
                                     "timed": "construct + iterate + reduce to one sum (hashjoin_bench.cc:126-133), best of the repetitions; the count and the ordered FNV come from a second, untimed walk over the same join",
                                     "checked": "count / sum / ordered FNV of the pairs == tests/golden strgen_join (the compiled reference's output)"}
    else:
        out["configs0_strgen_1M_ms"] = None
        out["configs0_strgen_1M"] = {"error": "tests/cpp/strgen_bench is not built (make)"}
    # small build sides (the reference's BM_hash_join_raw formulation, hashjoin_bench.cc:29-63): one global table, probe side unpartitioned
    n = 1 << 26
    for lb in (16, 20):
        R, S = ex.gen_build(1 << lb), ex.gen_uniform_domain(n, 1 << lb)
        ms, r = timed(lambda: ex.join_device(R, S, 0))
        assert int(r.n_matches) == n
        out["small_build_2p%d_x_2p26_count_ms" % lb] = ms
        roof("small_build_2p%d_x_2p26_count_ms" % lb, ms, 1 << lb, n)
        if lb == 16:
            msm, rm = timed(lambda: ex.join_device(R, S, H.HMJ_MATERIALIZE))
            assert int(rm.n_matches) == n
            out["small_build_2p16_x_2p26_materialize_ms"] = msm
            roof("small_build_2p16_x_2p26_materialize_ms", msm, 1 << lb, n, n)
            mso, ro = timed(lambda: ex.join_device(R, S, H.HMJ_ORDERED), reps=2)  # the operator's mode: rows in (key, rval, sval) order
            assert int(ro.n_matches) == n
            out["small_build_2p16_x_2p26_ordered_ms"] = mso
            roof("small_build_2p16_x_2p26_ordered_ms", mso, 1 << lb, n, n)
            out["small_build_2p16_ordered_path"] = ("sort on (key rank, payload) composites" if ex.last_timing()["path"] & H.HMJ_PATH_ORDER_BY_RANK_SORT
                                                    else "partitioned (%d bits)" % ex.last_timing()["radix_bits"])
            ex.release_result()
            ex.join_device(R, S, 0)
        tp = ex.last_timing()
        out["small_build_2p%d_path" % lb] = ("global table, probe side unpartitioned" if tp["path"] & H.HMJ_PATH_GLOBAL_TABLE else
                                              "one %d-bit slab pass over the probe side, its slabs probed in place" % tp["radix_bits"]
                                              if tp["path"] & H.HMJ_PATH_SLAB_ONE_PASS else "partitioned (%d bits)" % tp["radix_bits"])
        del R, S
    R, S = ex.gen_build(n), ex.gen_probe(n, n)
    ms, r = timed(lambda: ex.join_device(R, S, 0))
    assert int(r.n_matches) == n
    out["configs1_2p26_planner_bits%d_ms" % ex.last_timing()["radix_bits"]] = ms
    roof("configs1_2p26_planner_ms", ms, n, n)
    ex.set_radix_bits(10)
    try:
        ms, r = timed(lambda: ex.join_device(R, S, 0))
        assert int(r.n_matches) == n
        t = ex.last_timing()
        out["configs1_2p26_forced_10bit_ms"] = ms
        roof("configs1_2p26_forced_10bit_ms", ms, n, n)
        out["configs1_2p26_forced_10bit_plan"] = "%d passes, chunked LDS build (%d-row build partitions)" % (t["radix_passes"], n >> 10)
    finally:
        ex.set_radix_bits(None)
    # what a caller of the reference ctor experiences (hashjoin_bench.cc:126-133): HOST-resident relations in, ordered
    # rows back on the host (hmj_join_u64; pageable memory both ways).  PCIe-bound: 2 GiB up + 1.5 GiB down at 2^26.
    Bh, Ph = R.cpu().numpy().view(np.uint64), S.cpu().numpy().view(np.uint64)
    del R, S
    for name, fl in (("ordered", H.HMJ_ORDERED), ("count", 0)):
        ms, r = timed(lambda: ex.join_host(Bh, Ph, fl), reps=3)
        assert int(r.n_matches) == n
        out["host_entry_2p26_%s_ms" % name] = ms
    out["host_entry_pcie_floor_ms"] = "2 GiB up + 1.5 GiB down at the ~55 GB/s one hipMemcpy reaches here = 67 (ordered), 39 (count)"
    ex.release_result()
    del Bh, Ph
    n = 1 << 28
    R, S = ex.gen_build(n), ex.gen_probe(n, n)
    for name, fl in (("materialize", H.HMJ_MATERIALIZE), ("ordered", H.HMJ_ORDERED)):
        ms, r = timed(lambda: ex.join_device(R, S, fl), reps=2)
        assert int(r.n_matches) == n
        out["configs2_2p28_%s_ms" % name] = ms
        roof("configs2_2p28_%s_ms" % name, ms, n, n, n)
    ex.release_result()
    del S
    # the same build side as the dimension table of a foreign-key join: 2^24 keys, 2^28 probe rows, ordered rows
    Rf = ex.gen_build(1 << 24)
    Sf = ex.gen_uniform_domain(1 << 28, 1 << 24)
    ms, r = timed(lambda: ex.join_device(Rf, Sf, H.HMJ_ORDERED), reps=2)
    assert int(r.n_matches) == 1 << 28
    out["fk_2p24_x_2p28_ordered_ms"] = ms
    roof("fk_2p24_x_2p28_ordered_ms", ms, 1 << 24, 1 << 28, 1 << 28)
    ex.release_result()
    # ... and with runs of 64 probe rows per key (the run ranking is linear in the run length: DESIGN section 6)
    del Rf, Sf
    Rf = ex.gen_build(1 << 22)
    Sf = ex.gen_uniform_domain(1 << 28, 1 << 22)
    ms, r = timed(lambda: ex.join_device(Rf, Sf, H.HMJ_ORDERED), reps=2)
    assert int(r.n_matches) == 1 << 28
    out["fk_2p22_x_2p28_ordered_ms"] = ms
    roof("fk_2p22_x_2p28_ordered_ms", ms, 1 << 22, 1 << 28, 1 << 28)
    ex.release_result()
    del R, Rf, Sf
    torch.cuda.empty_cache()
    # the full sort (hmj_sort_u64_device = radix_int_non_inplace<u64,u64>, radix_bench_par.cc:126-127): 2^28 rows of
    # uniform 64-bit keys, and ids 0 .. 2^28 - 1 in random order; sortedness and the payload sum are checked
    for name, mkrel in (("uniform", lambda: ex.gen_build(1 << 28)),
                        ("dense", lambda: torch.stack([torch.randperm(1 << 28, device="cuda"), torch.arange(1 << 28, device="cuda")], 1).contiguous())):
        a = mkrel()
        ms, o = timed(lambda: ex.sort_device(a), reps=2)
        k = o[:, 0]
        top = torch.iinfo(torch.int64).min  # (flips the sign bit: signed compare of the flipped keys = unsigned order)
        assert bool(((k[1:] ^ top) >= (k[:-1] ^ top)).all()) and int(o[:, 1].sum()) == int(a[:, 1].sum()), name
        out["sort_2p28_%s_keys_ms" % name] = ms
        roof("sort_2p28_%s_keys_ms" % name, ms, 1 << 28, 0, sort=True)
        tp = ex.last_timing()["path"]
        out["sort_2p28_%s_path" % name] = ("MSD: two slab passes on the top %d varying bits + an LDS sort per partition" % ex.last_timing()["radix_bits"] if tp & H._lib.HMJ_PATH_SORT_MSD
                                           else "chain of slab passes + compaction" if tp & H.HMJ_PATH_SLAB else "exact passes")
        del a, o, k
        torch.cuda.empty_cache()
    # duplicate keys on BOTH sides (outside the reference's domain: its iterator drops matches there): 2^24 rows per side
    # drawn from 2^21 keys, ~1.3 * 10^8 result rows in (key, rval, sval) order
    g = torch.Generator(device="cuda")
    g.manual_seed(6)
    nd = 1 << 24
    mk = lambda: torch.stack([torch.randint(0, nd // 8, (nd,), device="cuda", generator=g, dtype=torch.int64) * 0x9E3779B97F4A7C15 % (1 << 62),
                              torch.arange(nd, device="cuda", dtype=torch.int64)], 1).contiguous()
    Rd, Sd = mk(), mk()
    ms, r = timed(lambda: ex.join_device(Rd, Sd, 0), reps=2)
    want_n = int(r.n_matches)
    ms, r = timed(lambda: ex.join_device(Rd, Sd, H.HMJ_ORDERED), reps=2)
    assert int(r.n_matches) == want_n
    out["dup8_ordered_ms"] = ms
    roof("dup8_ordered_ms", ms, nd, nd, want_n)
    out["dup8_rows"] = want_n
    ex.release_result()
    del Rd, Sd
    torch.cuda.empty_cache()
    # configs[4]: Zipf(0.9) build side of 2^24 rows over 2^24 distinct values, probe 2^30 uniform over the domain
    nb, npb, theta = 1 << 24, 1 << 30, 0.9
    w = 1.0 / np.arange(1, nb + 1, dtype=np.float64) ** theta
    cdf = np.cumsum(w) / w.sum()
    thr = np.empty(nb, np.uint64)
    big = cdf >= 1.0 - 2.0 ** -53
    thr[~big] = (cdf[~big] * 2.0 ** 64).astype(np.uint64)
    thr[big] = np.uint64((1 << 64) - 1)
    thr[-1] = np.uint64((1 << 64) - 1)
    thr_dev = torch.from_numpy(thr.view(np.int64).copy()).cuda()
    Rz = ex.gen_from_cdf(nb, thr_dev)
    Sz = ex.gen_uniform_domain(npb, nb)
    ms, r = timed(lambda: ex.join_device(Rz, Sz, 0), reps=2)
    out["configs4_zipf_2p24_x_2p30_count_ms"] = ms
    roof("configs4_zipf_2p24_x_2p30_count_ms", ms, nb, npb)
    out["configs4_matches"] = int(r.n_matches)
    cross = {k: int(getattr(r, k)) for k in ("n_matches", "sum_r", "sum_s")}
    ms, r = timed(lambda: ex.join_device(Rz, Sz, H.HMJ_FIRST_WINS | H.HMJ_SUM_PROBE), reps=2)
    out["configs4_first_wins_ms"] = ms
    roof("configs4_first_wins_ms", ms, nb, npb)
    # checked, not just printed: counts and sums recomputed in the generators' rank domain with plain torch ops
    # (tools/closed_forms.py; independent of the join kernels), cross product and first-wins (partitioned_hash.h:166-170)
    from tools.closed_forms import config5_checks

    want = config5_checks(torch, nb, npb, nb, thr_dev)
    assert cross == want["cross"], ("configs[4] cross product", cross, want["cross"])
    fwv = {k: int(getattr(r, k)) for k in ("n_matches", "sum_r", "sum_s")}
    assert fwv == want["first_wins"] and int(r.sum_probe_all) == want["sum_probe_all"], ("configs[4] first-wins", fwv, want)
    out["configs4_checked"] = "n_matches / sum_r / sum_s of both modes == torch rank-domain closed forms"
    del Rz, Sz
    torch.cuda.empty_cache()
    # (last only because it regrows the partition buffers hmj_reserve placed for the 2^28-row runs; its plan does not depend on
    #  what ran before it: what a context learns belongs to a workload, hmj_last_plan)
    # the reference's own sweep goes on to 10^9 rows (hashjoin_bench.cc:269-283): 5 * 10^8 x 5 * 10^8, count mode, on the
    # 17-bit plan the histogram-free slab partitioning makes since round 4 (9-bit + 8-bit pass)
    n = 500000000
    R, S = ex.gen_build(n), ex.gen_probe(n, n)
    ms, r = timed(lambda: ex.join_device(R, S, 0), reps=2)
    assert int(r.n_matches) == n and int(r.sum_r) == (n * (n - 1) // 2) % (1 << 64)
    t = ex.last_timing()
    out["scale_5e8_count_ms"] = ms
    roof("scale_5e8_count_ms", ms, n, n)
    out["scale_5e8_plan"] = "%d bits, %s" % (t["radix_bits"], "slab path" if t["path"] & H.HMJ_PATH_SLAB else "exact path")
    del R, S
    torch.cuda.empty_cache()
    out["roofline"] = dict(roofs, _what="per entry: algorithmic minimum bytes of the call (16 B per input row read once + 24 B per result row "
                                        "written once; sorts: 32 B per row), wall-clock ms of the whole call, fraction of the %.0f GB/s peak; "
                                        "per-kernel durations and PMC counters of the same workloads: profiles/r05a_paths_kernel_trace_and_pmc.txt" % HBM_PEAK_GBS)
    return out


METRIC = "probe-side tuples/s + achieved HBM GB/s, |R|=|S|=2^28 u64 keys"


def error_line(n_gpus, steps, warmup, what, detail=None):
    """The line a failed run prints instead of a result: same metric / unit, value null, "error" set."""
    d = {"metric": METRIC, "value": None, "unit": "probe tuples/s", "n_gpus": n_gpus, "steps": steps, "warmup": warmup,
         "higher_is_better": True, "error": what}
    if detail:
        d["detail"] = detail
    return json.dumps(d)


def self_launch(n_ranks, timeout_s, argv=None, steps=0, warmup=0):
    """`python bench.py --gpus N` without a launcher (WORLD_SIZE unset): this parent starts N fresh child processes
    -- one rank each, the same command line, RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set -- relays rank 0's JSON line
    and returns the worst child status.  The parent never touches the GPU (no torch import, no HIP call): every
    rank initialises its GPU in a process of its own, as under torch.distributed.run.  The reference reaches all
    of its workers from one call the same way (hashjoin.h:56-68 -> radix_hash.h:375-405).
    Bounded: after timeout_s of wall clock (or 30 s after the first child failed) every child still running is stopped
    by its own handle, each rank's stderr tail is relayed, and -- when rank 0 printed no line -- the parent prints one
    with "error" set.  argv: the child command (tests pass a stand-in)."""
    import socket
    import subprocess
    import tempfile
    import threading

    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = argv if argv is not None else [sys.executable, os.path.abspath(__file__)] + sys.argv[1:]
    logdir = tempfile.mkdtemp(prefix="hmj_bench_")
    procs, errs = [], []
    for r in range(n_ranks):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n_ranks), LOCAL_WORLD_SIZE=str(n_ranks),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HMJ_BENCH_SELF_LAUNCHED="1")
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        errs.append(open(os.path.join(logdir, "rank%d.stderr" % r), "w+b"))
        procs.append(subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE if r == 0 else errs[r], stderr=errs[r]))
    out0 = []
    reader = threading.Thread(target=lambda: out0.append(procs[0].stdout.read()), daemon=True)
    reader.start()  # rank 0 prints the line; EOF when it exits
    t_start, failed_at, why = time.time(), None, None
    while any(p.poll() is None for p in procs):
        now = time.time()
        if failed_at is None and any(p.poll() not in (None, 0) for p in procs):
            failed_at = now
        if why is None and now - t_start > timeout_s:
            why = "wall-clock limit of %.0f s reached with rank(s) %s still running" % (
                timeout_s, ",".join(str(i) for i, p in enumerate(procs) if p.poll() is None))
        elif why is None and failed_at is not None and now - failed_at > float(os.environ.get("HMJ_BENCH_GRACE_S", "30")):
            why = "rank(s) %s failed; the others were stopped %s s later" % (
                ",".join(str(i) for i, p in enumerate(procs) if p.poll() not in (None, 0)), os.environ.get("HMJ_BENCH_GRACE_S", "30"))
        if why is not None:
            for p in procs:  # each child by its own handle
                if p.poll() is None:
                    p.kill()
        time.sleep(0.05)
    reader.join(timeout=10.0)
    worst = 0
    for p in procs:
        rc = p.returncode
        if rc != 0 and (worst == 0 or abs(rc) > abs(worst)):
            worst = rc
    text = b"".join(out0).decode(errors="replace")
    sys.stdout.write(text)
    if worst != 0 or why is not None:
        tails = {}
        for r, f in enumerate(errs):
            f.seek(0)
            tail = f.read().decode(errors="replace")[-1500:]
            tails["rank%d" % r] = {"returncode": procs[r].returncode, "stderr_tail": tail}
            sys.stderr.write("---- rank %d (exit %s) stderr tail ----\n%s\n" % (r, procs[r].returncode, tail))
        if not any(l.startswith("{") for l in text.splitlines()):
            print(error_line(n_ranks, steps, warmup, why or "a rank exited with status %d" % worst,
                             {k: v["returncode"] for k, v in tails.items()}))
        if worst == 0:
            worst = 124
    else:
        for r, f in enumerate(errs):  # warnings of a good run stay visible
            f.seek(0)
            sys.stderr.write(f.read().decode(errors="replace"))
    for f in errs:
        f.close()
    sys.stdout.flush()
    return worst if worst >= 0 else 128 - worst


def measure_traffic_live(log2n, timeout_s=150.0):
    """HBM bytes per launch of the scatter kernels from rocprofv3 PMC counters, measured NOW: two child runs of this file
    (`rocprofv3 --pmc FETCH_SIZE -- python3 bench.py --steps 1 ...`, then WRITE_SIZE; counters in their own passes, never with
    a trace domain, the program itself after `--`), corrected as MI355X_MICROARCH.md prescribes for gfx950:
    bytes = (2 x FETCH_SIZE + WRITE_SIZE) x 1024.  Returns (bytes per launch averaged over slab A and slab B, detail) or
    (None, reason).  The parent keeps its buffers while the children run (2 x 26 GB of 288)."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile

    exe = shutil.which("rocprofv3") or ("/opt/rocm/bin/rocprofv3" if os.path.exists("/opt/rocm/bin/rocprofv3") else None)
    if exe is None:
        return None, "rocprofv3 not found"
    per_kernel = {}
    t_end = time.time() + timeout_s
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        d = tempfile.mkdtemp(prefix="hmj_pmc_")
        cmd = [exe, "--pmc", counter, "--output-format", "csv", "-d", d, "-o", "pmc", "--", sys.executable, os.path.abspath(__file__),
               "--steps", "1", "--warmup", "1", "--no-cpu", "--no-extra", "--no-traffic", "--log2n", str(log2n), "--timeout-s", "120"]
        env = dict(os.environ, TMPDIR="/tmp", HMJ_BENCH_SELF_LAUNCHED="1")
        try:
            left = t_end - time.time()
            if left < 10:
                return None, "out of time before the %s pass" % counter
            # (its own process group: past the limit the profiler AND the program under it are stopped, by that handle)
            p = subprocess.Popen(cmd, env=env, cwd="/tmp", stdout=subprocess.PIPE, stderr=subprocess.PIPE, start_new_session=True)
            try:
                _, perr = p.communicate(timeout=left)
            except subprocess.TimeoutExpired:
                try:
                    os.killpg(p.pid, 9)
                except OSError:
                    pass
                p.communicate()
                return None, "%s pass did not finish in time" % counter
            if p.returncode != 0:
                return None, "%s pass exited %d: %s" % (counter, p.returncode, perr.decode(errors="replace")[-200:])
            files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
            if not files:
                return None, "%s pass wrote no counter file" % counter
            for f in files:
                for row in csv.DictReader(open(f)):
                    k = row.get("Kernel_Name", "")
                    name = "slab_a" if "radix_slab_a_kernel" in k else "slab_b" if "radix_slab_b_kernel" in k else None
                    if name and row.get("Counter_Name") == counter:
                        per_kernel.setdefault(name, {}).setdefault(counter, []).append(float(row["Counter_Value"]))
        except Exception as e:  # (a measurement beside the line: never the reason a run fails)
            return None, "%s pass: %r" % (counter, e)
        finally:
            shutil.rmtree(d, ignore_errors=True)
    out = {}
    for name, cs in per_kernel.items():
        if "FETCH_SIZE" in cs and "WRITE_SIZE" in cs:
            f = sum(cs["FETCH_SIZE"]) / len(cs["FETCH_SIZE"])
            w = sum(cs["WRITE_SIZE"]) / len(cs["WRITE_SIZE"])
            out[name] = {"FETCH_SIZE_KB": round(f, 1), "WRITE_SIZE_KB": round(w, 1), "hbm_bytes_per_launch": int((2.0 * f + w) * 1024.0),
                         "launches": len(cs["FETCH_SIZE"])}
    if "slab_a" not in out or "slab_b" not in out:
        return None, "the counter files name no slab kernels"
    return (out["slab_a"]["hbm_bytes_per_launch"] + out["slab_b"]["hbm_bytes_per_launch"]) // 2, out


def arm_process_watchdog(timeout_s, rank, n_gpus, steps, warmup):
    """Under ANY launcher (torch.distributed.run included) a rank that is stuck -- a rendezvous, a barrier, a driver
    call -- must not keep the whole job alive: after timeout_s this thread prints the error line (rank 0) and ends
    the process with status 124.  The exchange itself gives up much earlier (hmj_comm_set_timeout_ms)."""
    import threading

    def fire():
        msg = "rank %d: wall-clock limit of %.0f s reached" % (rank, timeout_s)
        if rank == 0:
            print(error_line(n_gpus, steps, warmup, msg), flush=True)
        sys.stderr.write("bench.py: " + msg + "\n")
        sys.stderr.flush()
        os._exit(124)

    t = threading.Timer(timeout_s, fire)
    t.daemon = True
    t.start()
    return t


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--log2n", type=int, default=28, help="rows per relation PER GPU (log2)")
    ap.add_argument("--materialize", action="store_true", help="also write the (key,rval,sval) columns")
    ap.add_argument("--cpu-log2n", type=int, default=24)
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the other configs / modes timed beside the headline")
    ap.add_argument("--bits", type=int, default=-1, help="force total radix bits")
    ap.add_argument("--no-traffic", action="store_true",
                    help="do not measure roofline.traffic with rocprofv3 PMC passes of two short child runs (N = 1 only; ~40 s)")
    ap.add_argument("--timeout-s", type=float, default=900.0,
                    help="wall-clock limit of the whole run; past it every rank is stopped and an error line is printed")
    ap.add_argument("--step-timeout-s", type=float, default=120.0,
                    help="N > 1: deadline of one exchange step inside the library (hmj_comm_set_timeout_ms)")
    a = ap.parse_args()

    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(a.gpus, a.timeout_s, steps=a.steps, warmup=a.warmup))  # before anything touches the GPU
    watchdog = arm_process_watchdog(a.timeout_s, int(os.environ.get("RANK", "0")), a.gpus, a.steps, a.warmup)
    # (multi-process GPU work on this image needs dmabuf IPC: RCCL fails with hipIpcGetMemHandle otherwise; set before HIP starts)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

    import torch
    import torch.distributed as dist

    import hashmergejoin_amd as H
    from hashmergejoin_amd import dist as hdist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    force_dist = os.environ.get("HMJ_FORCE_DIST") == "1"  # dev: run the exchange path with 1 rank (RCCL self send/recv)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if a.gpus != world and rank == 0:
        print("bench.py: --gpus %d but WORLD_SIZE=%d: running %d ranks" % (a.gpus, world, world), file=sys.stderr)
    ndev = torch.cuda.device_count()  # (counting devices does not initialise the GPU)
    # RCCL refuses two ranks on one device: with fewer GPUs than ranks (rehearsals on a one-GPU box) the ranks share
    # the GPUs and the exchange runs over the library's callback transport on gloo; HMJ_DIST_BACKEND overrides
    backend = os.environ.get("HMJ_DIST_BACKEND", "nccl" if ndev >= world else "gloo")
    local_dev = local_rank % max(ndev, 1)
    torch.cuda.set_device(local_dev)
    dev = torch.device("cuda", local_dev)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    n = 1 << a.log2n           # rows per relation on this GPU
    n_total = n * world        # |R| = |S| of the whole job
    ex = H.Executor(local_dev)
    if a.bits >= 0:
        ex.set_radix_bits(a.bits)
    if world > 1:
        hdist.init_comm(ex, timeout_s=a.step_timeout_s)  # RCCL communicator inside the library (or the gloo callback transport)
    elif force_dist:
        hdist.init_comm_single(ex, timeout_s=a.step_timeout_s)
    # test hook (tests/test_dist_gpu.py): "kill:<rank>:<step>" ends that rank before its step, "stall:<rank>:<step>" makes
    # it sleep instead of entering the step -- the other ranks must come back with HMJ_E_TIMEOUT and the run must end
    fault = os.environ.get("HMJ_BENCH_FAULT", "").split(":")
    fault = (fault[0], int(fault[1]), int(fault[2])) if len(fault) == 3 else None
    step_no = [0]
    distributed = world > 1 or force_dist
    # synthetic relations generated on device: this rank's row shard [rank*n, (rank+1)*n)
    R = ex.gen_build(n, start=rank * n)
    S = ex.gen_probe(n, n_total, start=rank * n)
    flags = H.HMJ_MATERIALIZE if a.materialize else 0
    ex.set_profiling(True)

    def step():
        if not distributed:
            res = ex.join_device(R, S, flags)
            return res, ex.last_timing(), None
        if fault and fault[1] == rank and fault[2] == step_no[0]:
            if fault[0] == "kill":
                os.kill(os.getpid(), 9)
            time.sleep(10 * a.step_timeout_s)
        step_no[0] += 1
        try:
            loc, glob = ex.exchange_join(R, S, flags)  # owner split, exchange rounds, prepared build, local join
        except H.HmjError as e:
            # a step that failed (HMJ_E_TIMEOUT: a peer never took part) ends the run on this rank at once: the line says why
            if rank == 0:
                print(error_line(world, a.steps, a.warmup, "exchange step %d failed: %s" % (step_no[0] - 1, e)), flush=True)
            sys.stderr.write("bench.py rank %d: exchange step failed: %s\n" % (rank, e))
            sys.stderr.flush()
            os._exit(5)  # (no teardown: the communicator is gone and a torch barrier would wait for the lost peer)
        return glob, ex.last_timing(), ex.last_exchange_info()

    # Workspace ahead of the first join, as a caller who cares about the first join's latency does (hmj_reserve):
    # this is also where the library may search for well-placed partition buffers, under its wall-clock budget
    # (`placement`; a join that allocates on its own only probes what it got).  Outside the timed region.
    reserve_ms = None
    malloc_cost = None
    if not distributed:
        torch.cuda.synchronize()
        malloc_cost = hipmalloc_cost_ms()
        t_w = time.perf_counter()
        ex.reserve(n, n, n if a.materialize else 0, flags)
        torch.cuda.synchronize()
        reserve_ms = (time.perf_counter() - t_w) * 1e3
    first_ms = None
    for i in range(a.warmup):
        t_w = time.perf_counter()
        res, _, _ = step()
        if i == 0:
            torch.cuda.synchronize()
            first_ms = (time.perf_counter() - t_w) * 1e3  # first join of the context (N > 1: it creates the workspace)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    agg, xagg = {}, {}
    t0 = time.perf_counter()
    for _ in range(a.steps):
        res, tm, xi = step()
        for k, v in tm.items():
            if k.startswith(("ms_", "bytes_", "n_scatter")):
                agg[k] = agg.get(k, 0) + v
        for k, v in (xi or {}).items():
            if k.startswith("ms_"):
                xagg[k] = xagg.get(k, 0) + v
        last_tm, last_xi = tm, xi
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    n_matches = int(res.n_matches)  # distributed: the reduction over all ranks (hmj_exchange_join's global_out)
    probe_ms_slowest = agg.get("ms_probe_count", 0.0) / max(1, a.steps)
    if world > 1:
        rdev = dev if backend == "nccl" else torch.device("cpu")
        # slowest rank: whole-step wall time, and its build+probe kernel time per step (the probe-phase rate below)
        t = torch.tensor([dt, probe_ms_slowest], dtype=torch.float64, device=rdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt, probe_ms_slowest = float(t[0].item()), float(t[1].item())
    assert n_matches == n_total, "every probe row must match exactly once (%d != %d)" % (n_matches, n_total)

    exit_code = 0
    if rank == 0:
        K = a.steps
        ms_step = dt / K * 1e3
        # dominant kernels by time: the radix scatters (2 relations x 2 passes per join)
        launches = max(1, agg.get("n_scatter_launches", 0))
        sc_ms = agg.get("ms_scatter", 0.0) / launches
        sc_bytes = agg.get("bytes_scatter", 0) / launches  # 32 B per row: 16 read + 16 written
        pr_ms = agg.get("ms_probe_count", 0.0) / K
        pr_bytes = agg.get("bytes_probe_count", 0) / K      # 16*(n_build + n_probe), read only
        slab = bool(last_tm["path"] & H.HMJ_PATH_SLAB)
        traffic, traffic_note = None, "not collected in this run (PMC passes need rocprofv3)"
        tj = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(tj) and a.log2n == 28 and slab:
            try:
                tjs = json.load(open(tj))
                traffic = tjs.get("radix_scatter_kernel", {}).get("hbm_bytes_per_launch")
                traffic_note = "from %s (rocprofv3 --pmc passes of this workload, %s), NOT measured by this run" % (
                    "profiles/pmc_traffic.json", tjs.get("_round", "round 1"))
            except Exception:
                traffic = None
        traffic_detail = None
        # (not under a profiler: a run that is itself being profiled -- tools/profile_gpu.sh, anybody's rocprofv3 -- starts none)
        profiled = "rocprof" in os.environ.get("LD_PRELOAD", "").lower() or any("ROCPROF" in k or k.startswith("ROCP_") for k in os.environ)
        if world == 1 and not distributed and not a.no_traffic and slab and not profiled and os.environ.get("HMJ_BENCH_TRAFFIC", "1") != "0":
            live, detail = measure_traffic_live(a.log2n)
            if live is not None:
                traffic, traffic_detail = live, detail
                traffic_note = ("measured by this run: rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE passes of two child runs of "
                                "`bench.py --steps 1 --warmup 1 --no-cpu --no-extra`, bytes = (2 x FETCH_SIZE + WRITE_SIZE) x 1024 "
                                "(MI355X_MICROARCH.md, gfx950), mean over the launches of slab A and slab B")
            else:
                traffic_note = "live PMC passes failed (%s); %s" % (detail, traffic_note)

        def roof(nbytes, ms, tr=None):
            ach = (nbytes / (ms * 1e-3) / 1e9) if ms > 0 else 0.0
            return {"bound": "hbm", "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": tr,
                    "bytes_per_launch": int(nbytes), "ms_per_launch": round(ms, 4)}

        # what a plain device-to-device copy reaches on this box (read + write bytes), for scale beside
        # the spec peak: the scatter is a copy with a permutation
        copy_gbs = None
        try:
            src = torch.empty(1 << 28, dtype=torch.int64, device=dev)  # 2 GiB
            dst = torch.empty_like(src)
            dst.copy_(src)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                dst.copy_(src)
            e1.record()
            torch.cuda.synchronize()
            copy_gbs = round(2 * src.numel() * 8 * 5 / (e0.elapsed_time(e1) * 1e-3) / 1e9, 1)
            del src, dst
        except Exception:
            copy_gbs = None

        pa = agg.get("ms_scatter_pass0", 0.0) / max(1, launches // 2)
        pb = agg.get("ms_scatter_pass1", 0.0) / max(1, launches // 2)
        rows_launch = sc_bytes / 32.0
        if slab:
            sc_kernel = "radix_slab_a_kernel<512,256> + radix_slab_b_kernel<512,256>"
            sc_desc = "write-combining stable scatter into private slabs, no histogram; mean over the launches of a join (2 x pass A + 2 x pass B)"
            pr_kernel = "probe_count_fast_kernel<1024,13,PCOUNT=false,SLAB=true,OUT=0>"
        else:
            sc_kernel = "radix_scatter_wc_kernel<512,256>"
            sc_desc = "write-combining stable scatter with histogram offsets; mean over the launches of a join"
            pr_kernel = "probe_count_fast_kernel (dense layout) + probe_kernel<0> for set-aside partitions"
        if distributed and last_xi["owner_mode"] == 3:
            # the digit path: one first pass per relation on the shard (radix_scatter_wc_kernel, its digit = the owner)
            # + the remaining passes of every round's join; launch counts differ per pass, so no A / B split here
            sc_kernel = "radix_scatter_wc_kernel<512,256> (first pass = owner) + " + sc_kernel + " (per-round joins)"
            sc_desc = "mean over all scatter launches of a step (32 B per row and launch)"
            pa = pb = 0.0
            digit_bits = last_xi["digit_bits"]
            plan_txt = "%d-bit first pass (owner) + %d-bit radix in %d LSD passes per round" % (digit_bits, last_tm["radix_bits"], last_tm["radix_passes"])
        else:
            plan_txt = "%d-bit radix in %d LSD passes" % (last_tm["radix_bits"], last_tm["radix_passes"])
        line = {
            "metric": METRIC,
            "value": n_total * K / dt,
            "unit": "probe tuples/s",
            "n_gpus": world, "steps": K, "warmup": a.warmup, "ms_per_step": ms_step,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u64", "data": "synthetic",
            "config": {"workload": "|R|=|S|=2^%d x %d GPU: u64 key / 8 B payload, 100%% match, %s, "
                                   "LDS build+probe, %s" % (a.log2n, world, plan_txt,
                                                            "materialised columns" if a.materialize else "count+sum (hashjoin_bench.cc:131-133)"),
                       "rows_per_relation_per_gpu": n, "rows_per_relation_total": n_total,
                       "parallelism": "radix-sharded x%d" % world},
            "roofline": dict(roof(sc_bytes, sc_ms, traffic), kernel=sc_kernel, what=sc_desc, traffic_source=traffic_note, traffic_detail=traffic_detail,
                             device_copy_GBps=copy_gbs, launches_per_step=launches // K,
                             pass_a={"ms_per_launch": round(pa, 4), "GBps": round(32.0 * rows_launch / (pa * 1e-3) / 1e9, 1) if pa > 0 else None},
                             pass_b={"ms_per_launch": round(pb, 4), "GBps": round(32.0 * rows_launch / (pb * 1e-3) / 1e9, 1) if pb > 0 else None}),
            "roofline_probe": dict(roof(pr_bytes, pr_ms), kernel=pr_kernel,
                                   probe_tuples_per_s=round(last_tm["bytes_probe_count"] / 32.0 / (pr_ms * 1e-3)) if pr_ms > 0 else None),
            "phases_ms_per_step": {k[3:]: round(v / K, 4) for k, v in agg.items() if k.startswith("ms_")},
            # why the scatter passes ran at the rate they did on THIS box: the fill rate of each partition buffer the
            # library kept (write bandwidth is a property of the physical memory behind a buffer)
            "placement": {"probing": os.environ.get("HMJ_PLACE", "default: joins probe only; hmj_reserve searches (<= 4 candidates, "
                                                                   "<= HMJ_PLACE_BUDGET_MS = 50 ms per buffer)"),
                          "buffers": ex.placement_info(),
                          "hipmalloc_cost": malloc_cost,
                          "reserve_ms": None if reserve_ms is None else round(reserve_ms, 2),
                          "first_join_ms": None if first_ms is None else round(first_ms, 2)},
        }
        if distributed:
            transport = "rccl" if (backend == "nccl" or force_dist) else "callbacks over " + backend
            owner = {0: "none (one rank: plain local join)", 1: "hash of the key (fallback: separate owner split)",
                     2: "key ranges between sample quantiles", 3: "ranges of the first radix pass's digit"}[last_xi["owner_mode"]]
            line["exchange"] = dict({k[3:]: round(v / K, 3) for k, v in xagg.items()}, unit="ms per step on rank 0",
                                    n_ranks=last_xi["n_ranks"], rccl_ranks=last_xi["n_ranks"] if transport == "rccl" else 0,
                                    transport=transport, owner=owner, digit_bits=last_xi["digit_bits"],
                                    digit_low=last_xi["digit_low"], rounds_build=last_xi["rounds_build"],
                                    rounds_probe=last_xi["rounds_probe"], joins_per_step=last_xi["n_subjoins"],
                                    sample_max_share=round(last_xi["sample_max_share"], 3),
                                    recv_rows_rank0=[last_xi["recv_build"], last_xi["recv_probe"]],
                                    what={"split": "first radix pass (or owner split) of both shards + count read-back, host clock",
                                          "exchange_build / exchange_probe": "rounds on the communication stream (HIP events)",
                                          "kernels": "this rank's own kernels: pre-pass + per-round joins (HIP events)",
                                          "exposed": "total - kernels: what the exchange and its synchronisation add to a step"})
            # the probe phase alone, aggregated over the ranks: all probe rows / the slowest rank's build+probe kernel time
            pb = last_tm["bytes_probe_count"]
            if probe_ms_slowest > 0:
                line["probe_phase"] = {"probe_tuples_per_s_all_ranks": round(n_total / (probe_ms_slowest * 1e-3)),
                                       "slowest_rank_ms_per_step": round(probe_ms_slowest, 4),
                                       "per_gpu": roof(pb, probe_ms_slowest)}
        if world == 1 and not distributed:
            del R, S
            torch.cuda.empty_cache()
            if not a.no_extra:
                # the extra runs CHECK their results (match counts, configs[4] against closed forms): a failure there
                # is a wrong join, so the line says so and the process exits non-zero after printing it
                try:
                    line["extra"] = extra_runs(ex, H, torch)
                    line["extra_ok"] = True
                except Exception as e:
                    line["extra"] = {"error": repr(e)}
                    line["extra_ok"] = False
                    exit_code = 3
            if not a.no_cpu:
                try:
                    cores, aff = usable_cores()
                    line["cpu_baseline"] = cpu_baseline(a.cpu_log2n, cores, aff)
                except Exception as e:  # the baseline is reporting only; never fail the bench on it
                    line["cpu_baseline"] = {"error": repr(e)}
        print(json.dumps(line), flush=True)
    watchdog.cancel()
    ex.close()
    if world > 1:
        dist.destroy_process_group()
    if exit_code:
        sys.exit(exit_code)


if __name__ == "__main__":
    main()
