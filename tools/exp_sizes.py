#!/usr/bin/env python3
"""Count-mode join time vs size (wall per call, and the GPU phases inside)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import hashmergejoin_amd as H
ex = H.Executor(0)
for n in [1000, 10**4, 10**5, 10**6, 1 << 22, 1 << 24, 1 << 26]:
    R, S = ex.gen_build(n), ex.gen_probe(n, n)
    for fl, name in [(0, "count"), (H.HMJ_ORDERED, "ordered")]:
        ex.set_profiling(False)
        for _ in range(3): ex.join_device(R, S, fl)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        reps = 20
        for _ in range(reps): ex.join_device(R, S, fl)
        torch.cuda.synchronize(); wall = (time.perf_counter() - t0) / reps * 1e3
        ex.set_profiling(True); ex.join_device(R, S, fl); t = ex.last_timing()
        gpu = sum(t[k] for k in ("ms_hist", "ms_scan", "ms_scatter", "ms_offsets", "ms_probe_count", "ms_out_scan", "ms_probe_write", "ms_order"))
        print("n=%9d %-8s wall %.3f ms  (kernels %.3f ms, bits %d) -> %.1f M tuples/s" % (n, name, wall, gpu, t["radix_bits"], n / wall / 1e3), flush=True)
