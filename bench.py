#!/usr/bin/env python3
"""bench.py -- probe-side tuples/s of the radix-partitioned hash join on MI355X.

A "step" is one whole pass of the hot path over one batch of synthetic relations that are already
resident in HBM: radix-partition R, radix-partition S, bucket-local build + probe in count/sum mode
(the reduction hashjoin_bench.cc:131-133 performs).  value = probe rows of all ranks x steps / time.

N=1 workload: BASELINE.json configs[2]: |R|=|S|=2^28 u64 key / 8 B payload, 100 % match.
N>1: weak scaling, 2^28 rows per relation per GPU (N=8 -> |R|=|S|=2^31, configs[3]); each step adds
the owner split + RCCL all-to-all exchange of both relations.

Prints ONE JSON line (rank 0).  Extra objects: "roofline" (dominant kernel, HIP-event timed inside
the timed region), "roofline_probe" (the build+probe kernel), "cpu_baseline" (the compiled reference
timed on the host cores, N=1 only).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def cpu_baseline(log2n, threads):
    """The reference's own pthread CPU path on a bounded sample of the same workload, timed on this
    box's host cores.  kind "reference": oracle/_ref/libhmj_ref.so is the real reference compiled
    from its own headers (oracle/Makefile).  Falls back to the single-thread C port."""
    import numpy as np

    from oracle.pyoracle import Oracle, Reference

    orc, ref = Oracle(), Reference()
    n = 1 << log2n
    B, P = orc.gen_build(n), orc.gen_probe(n, n)
    out = {"cores": threads, "unit": "probe tuples/s", "sample": "|R|=|S|=2^%d u64 key / 8 B payload, same generator, 100%% match" % log2n}
    if ref.available:
        rh, sh = ref.pairs_new(B), ref.pairs_new(P)
        best = None
        for _ in range(2):
            t0 = time.perf_counter()
            cnt, sm = ref.hashmergejoin_pairs(rh, sh, threads)  # ctor + iterate, hashjoin_bench.cc:126-133
            dt = time.perf_counter() - t0
            best = dt if best is None else min(best, dt)
        assert cnt == n
        out.update(kind="reference", value=n / best, seconds=best, what="HashMergeJoin ctor + iterate (hashjoin_bench.cc:126-133)")
        # what radix_bench_par times for u64 (radix_bench_par.cc:126-127)
        oh = ref.pairs_new(B)
        t0 = time.perf_counter()
        ref.radix_int_non_inplace_pairs(rh, oh, threads)
        dt = time.perf_counter() - t0
        out["radix_int_non_inplace_keys_per_s"] = n / dt
        m = min(n, 1 << 22)  # radix_bench_par.cc:96; the in-place sort is much slower, keep the sample short
        t0 = time.perf_counter()
        ref.radix_int_inplace(B[:m], threads)
        out["radix_int_inplace_keys_per_s"] = m / (time.perf_counter() - t0)
        t0 = time.perf_counter()
        psum, found = ref.partitioned_join_sum(P, B, threads, 10)  # hashjoin_bench.cc:88-96
        out["partitioned_build_probe_tuples_per_s"] = n / (time.perf_counter() - t0)
        for h in (rh, sh, oh):
            ref.pairs_free(h)
    else:
        t0 = time.perf_counter()
        cnt, sm, _ = orc.hashmergejoin(B, P, 1, cap=0)
        dt = time.perf_counter() - t0
        out.update(kind="port", cores=1, value=n / dt, seconds=dt, what="C restatement of HashMergeJoin ctor + iterate, 1 thread")
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--log2n", type=int, default=28, help="rows per relation PER GPU (log2)")
    ap.add_argument("--materialize", action="store_true", help="also write the (key,rval,sval) columns")
    ap.add_argument("--cpu-log2n", type=int, default=24)
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--bits", type=int, default=-1, help="force total radix bits")
    a = ap.parse_args()

    import torch
    import torch.distributed as dist

    import hashmergejoin_amd as H
    from hashmergejoin_amd import dist as hdist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    force_dist = os.environ.get("HMJ_FORCE_DIST") == "1"  # dev: run the exchange path with 1 rank
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if a.gpus != world:
        if world == 1 and a.gpus > 1:
            sys.exit("launch with torch.distributed.run --nproc-per-node %d for --gpus %d" % (a.gpus, a.gpus))
    backend = os.environ.get("HMJ_DIST_BACKEND", "nccl")  # "gloo": rehearse N ranks on one GPU
    ndev = torch.cuda.device_count()
    local_dev = local_rank % max(ndev, 1)
    torch.cuda.set_device(local_dev)
    dev = torch.device("cuda", local_dev)
    if world > 1 or force_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    n = 1 << a.log2n           # rows per relation on this GPU
    n_total = n * world        # |R| = |S| of the whole job
    ex = H.Executor(local_dev)
    if a.bits >= 0:
        ex.set_radix_bits(a.bits)
    # synthetic relations generated on device: this rank's row shard [rank*n, (rank+1)*n)
    R = ex.gen_build(n, start=rank * n)
    S = ex.gen_probe(n, n_total, start=rank * n)
    flags = H.HMJ_MATERIALIZE if a.materialize else 0
    ex.set_profiling(True)
    if world > 1:
        ex.set_key_prefix_bits(hdist.owner_bits(world))  # received rows share their top owner bits

    def step():
        if world == 1 and not force_dist:
            res = ex.join_device(R, S, flags)
            return res, ex.last_timing()
        b = max(hdist.owner_bits(world), 1 if force_dist else 0)
        if force_dist:  # one rank: everything is sent to self, as one message per relation
            recv = []
            for rel in (R, S):
                parted, off = ex.partition_device(rel, 64 - b, b)
                rows, _ = hdist.exchange_rows(parted, hdist.split_counts_from_offsets(off[[0, -1]]))
                recv.append(rows)
            res = ex.join_device(recv[0], recv[1], flags)
        else:  # owner split, RCCL all-to-all hidden behind the splits / build-side partitioning, join
            res = hdist.pipelined_exchange_join(ex, R, S, b, flags)
        return res, ex.last_timing()

    for _ in range(a.warmup):
        res, _ = step()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    agg = {}
    t0 = time.perf_counter()
    for _ in range(a.steps):
        res, tm = step()
        for k, v in tm.items():
            if k.startswith(("ms_", "bytes_", "n_scatter")):
                agg[k] = agg.get(k, 0) + v
        last_tm = tm
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    n_local = int(res.n_matches)
    if world > 1:
        rdev = dev if backend == "nccl" else torch.device("cpu")
        t = torch.tensor([dt], dtype=torch.float64, device=rdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        m = torch.tensor([n_local], dtype=torch.int64, device=rdev)
        dist.all_reduce(m)
        n_matches = int(m.item())
    else:
        n_matches = n_local
    assert n_matches == n_total, "every probe row must match exactly once (%d != %d)" % (n_matches, n_total)

    if rank == 0:
        K = a.steps
        ms_step = dt / K * 1e3
        # dominant kernel by time: the radix scatter (4 launches per join at 2 passes x 2 relations)
        launches = max(1, agg.get("n_scatter_launches", 0))
        sc_ms = agg.get("ms_scatter", 0.0) / launches
        sc_bytes = agg.get("bytes_scatter", 0) / launches  # 32 B per row: 16 read + 16 written
        pr_ms = agg.get("ms_probe_count", 0.0) / K
        pr_bytes = agg.get("bytes_probe_count", 0) / K      # 16*(n_build + n_probe), read only
        traffic = None
        tj = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(tj):
            try:
                traffic = json.load(open(tj)).get("radix_scatter_kernel", {}).get("hbm_bytes_per_launch")
            except Exception:
                traffic = None

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

        line = {
            "metric": "probe-side tuples/s + achieved HBM GB/s, |R|=|S|=2^28 u64 keys",
            "value": n_total * K / dt,
            "unit": "probe tuples/s",
            "n_gpus": world, "steps": K, "warmup": a.warmup, "ms_per_step": ms_step,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u64", "data": "synthetic",
            "config": {"workload": "|R|=|S|=2^%d x %d GPU: u64 key / 8 B payload, 100%% match, %d-bit radix in %d LSD passes, "
                                   "LDS build+probe, %s" % (a.log2n, world, last_tm["radix_bits"], last_tm["radix_passes"],
                                                            "materialised columns" if a.materialize else "count+sum (hashjoin_bench.cc:131-133)"),
                       "rows_per_relation_per_gpu": n, "rows_per_relation_total": n_total,
                       "parallelism": "radix-sharded x%d" % world},
            "roofline": dict(roof(sc_bytes, sc_ms, traffic), kernel="radix_scatter_kernel", device_copy_GBps=copy_gbs,
                             launches_per_step=launches // K, kernel_name="radix_slab_a_kernel / radix_slab_b_kernel (write-combining stable scatter, mean of the launches)"),
            "roofline_probe": dict(roof(pr_bytes, pr_ms), kernel="probe_kernel<count>",
                                   probe_tuples_per_s=round(n / (pr_ms * 1e-3)) if pr_ms > 0 else None),
            "phases_ms_per_step": {k[3:]: round(v / K, 4) for k, v in agg.items() if k.startswith("ms_")},
        }
        if world == 1 and not a.no_cpu:
            try:
                try:
                    cores = len(os.sched_getaffinity(0))
                except Exception:
                    cores = os.cpu_count() or 1
                cores = min(cores, int(os.environ.get("HMJ_CPU_THREADS", "16")))  # the box's CPU share for one GPU
                line["cpu_baseline"] = cpu_baseline(a.cpu_log2n, cores)
            except Exception as e:  # the baseline is reporting only; never fail the bench on it
                line["cpu_baseline"] = {"error": repr(e)}
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
