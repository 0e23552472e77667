#!/usr/bin/env python3
"""Per-kernel times of the headline join (slab A / slab B / probe) for one build of the library.
   HMJ_LIB=build/variants/libhmj_x.so python tools/exp_kernels.py [log2n] [reps]
With a -DHMJ_STAMPS build it also prints the phase breakdown of the write-combining scatter kernels."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401
import hashmergejoin_amd as H

log2n = int(sys.argv[1]) if len(sys.argv) > 1 else 28
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 6
flags = int(sys.argv[3]) if len(sys.argv) > 3 else 0
ex = H.Executor(0)
ex.set_profiling(True)
n = 1 << log2n
R, S = ex.gen_build(n), ex.gen_probe(n, n)
L = ex.L
stamps = getattr(L, "hmj_dev_stamps", None) if hasattr(L, "hmj_dev_stamps") else None
rows = []
for i in range(reps + 2):
    r = ex.join_device(R, S, flags)
    assert int(r.n_matches) == n or os.environ.get('HMJ_NOCHECK')
    t = ex.last_timing()
    if i >= 2:
        rows.append(t)
    elif stamps is not None and i == 1:
        buf = (C.c_uint64 * 32)()
        L.hmj_dev_stamps(buf, 1)  # drop the warm-up launches
best = lambda k: min(t[k] for t in rows)
mean = lambda k: sum(t[k] for t in rows) / len(rows)
tag = os.path.basename(os.environ.get("HMJ_LIB", "release"))
print("%-28s 2^%d total %.3f (best %.3f) | passA %.3f passB %.3f per launch | probe %.3f write %.3f order %.3f | path %#x" % (
    tag, log2n, mean("ms_total"), best("ms_total"), mean("ms_scatter_pass0") / 2, mean("ms_scatter_pass1") / 2,
    mean("ms_probe_count"), mean("ms_probe_write"), mean("ms_order"), rows[-1]["path"]), flush=True)
if stamps is not None:
    buf = (C.c_uint64 * 32)()
    L.hmj_dev_stamps(buf, 1)
    names = ["vmwait", "rank", "bar1", "plan", "bar2", "stage", "bar3", "copyout+keep", "bar4", "carry+clear", "bar5", None, "prefetch(rest)", "B:issue"]
    for base, kn in ((0, "pass A"), (16, "pass B")):
        tiles = max(1, buf[base + 11])
        tot = max(1, sum(buf[base + i] for i in range(14) if i != 11))
        print("   %s: stamps per wave-tile (shader cycles); wave-tiles = %d" % (kn, tiles))
        for i, nm in enumerate(names):
            if nm is None:
                continue
            print("     %-14s %8.0f  %5.1f %%" % (nm, buf[base + i] / tiles, 100.0 * buf[base + i] / tot))
        print("     %-14s %8.0f" % ("sum", tot / tiles))
