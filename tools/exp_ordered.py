#!/usr/bin/env python3
"""Per-kernel times of the ordered joins the C++ operator asks for (HMJ_ORDERED), for profiling runs:
   python tools/exp_ordered.py uniq [log2n] [reps]        unique keys both sides, |R| = |S| = 2^log2n (default 28)
   python tools/exp_ordered.py fk [log2b] [log2p] [reps]  foreign-key join: 2^log2b unique build keys, 2^log2p probe rows
Prints the phase times of the last join (HIP events) and the code path."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401
import hashmergejoin_amd as H

kind = sys.argv[1] if len(sys.argv) > 1 else "uniq"
ex = H.Executor(0)
ex.set_profiling(True)
if kind == "uniq":
    log2n = int(sys.argv[2]) if len(sys.argv) > 2 else 28
    reps = int(sys.argv[3]) if len(sys.argv) > 3 else 4
    n = 1 << log2n
    R, S = ex.gen_build(n), ex.gen_probe(n, n)
    want = n
else:
    log2b = int(sys.argv[2]) if len(sys.argv) > 2 else 24
    log2p = int(sys.argv[3]) if len(sys.argv) > 3 else 28
    reps = int(sys.argv[4]) if len(sys.argv) > 4 else 4
    R, S = ex.gen_build(1 << log2b), ex.gen_uniform_domain(1 << log2p, 1 << log2b)
    want = 1 << log2p
rows = []
for i in range(reps + 2):
    r = ex.join_device(R, S, H.HMJ_ORDERED)
    assert int(r.n_matches) == want
    if i >= 2:
        rows.append(ex.last_timing())
mean = lambda k: sum(t[k] for t in rows) / len(rows)
print("%s ordered: total %.3f ms | partition %.3f | write %.3f | order %.3f | scan %.3f | path %#x bits %d" % (
    " ".join(sys.argv[1:]) or "uniq", mean("ms_total"), mean("ms_partition_build") + mean("ms_partition_probe"), mean("ms_probe_write"),
    mean("ms_order"), mean("ms_out_scan"), rows[-1]["path"], rows[-1]["radix_bits"]), flush=True)
