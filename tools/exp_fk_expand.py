#!/usr/bin/env python3
"""Ordered foreign-key joins (unique build keys, f probe rows per key): the one-pass ordered write ranking inside whole runs
(HMJ_FK_PAYLOAD_BUCKETS=0, as until round 4) and inside (build rank, payload position) buckets (default, from fan-out 24 on),
the composite sort where it applies, and -- with EXP_WITH_EXPANSION=1 -- count + scan + ordered expansion
(HMJ_EXPAND_FK_FANOUT=1).  One executor at a time (their workspaces together do not fit the card at 2^28 rows).
usage: exp_fk_expand.py [log2 probe rows = 28]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import hashmergejoin_amd as H

P = int(sys.argv[1]) if len(sys.argv) > 1 else 28
CONFIGS = [("one-pass write, whole runs", {"HMJ_GTABLE_SORT": "0", "HMJ_FK_PAYLOAD_BUCKETS": "0"}),
           ("one-pass write", {"HMJ_GTABLE_SORT": "0"}),
           ("composite sort", {"HMJ_GTABLE_SORT_FANOUT": "1"})]
if os.environ.get("EXP_WITH_EXPANSION"):
    CONFIGS.insert(2, ("expansion", {"HMJ_GTABLE_SORT": "0", "HMJ_EXPAND_FK_FANOUT": "1"}))


def timed(e, R, S, fl, reps=3):
    for _ in range(2):
        r = e.join_device(R, S, fl)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        r = e.join_device(R, S, fl)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3, r


gen = H.Executor(0)
for k in range(P - 11, P - 2):
    R, S = gen.gen_build(1 << k), gen.gen_uniform_domain(1 << P, 1 << k)
    fl = H.HMJ_ORDERED | H.HMJ_CHECKSUM
    out, cks = [], []
    for name, env in CONFIGS:
        os.environ.update(env)
        try:
            e = H.Executor(0)
        finally:
            for v in env:
                del os.environ[v]
        e.set_profiling(True)
        m, r = timed(e, R, S, fl)
        t = e.last_timing()
        cks.append(r.checks())
        out.append("%s %.3f ms (b%d path %#x; part %.2f count %.2f write %.2f)" % (name, m, t["radix_bits"], t["path"], t["ms_partition_build"] + t["ms_partition_probe"], t["ms_probe_count"], t["ms_probe_write"]))
        e.close()
        torch.cuda.empty_cache()
    print("nb=2^%d np=2^%d fan-out %d | %s%s" % (k, P, 1 << (P - k), " | ".join(out), "" if all(c == cks[0] for c in cks) else " MISMATCH"), flush=True)
    del R, S
