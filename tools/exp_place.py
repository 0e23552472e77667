#!/usr/bin/env python3
"""First-join cost and steady state with / without the placement tuning of the slab buffers (HMJ_PLACE=0/1)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import hashmergejoin_amd as H
n = 1 << (int(sys.argv[1]) if len(sys.argv) > 1 else 28)
ex = H.Executor(0); ex.set_profiling(True)
R, S = ex.gen_build(n), ex.gen_probe(n, n)
torch.cuda.synchronize()
t0 = time.perf_counter(); ex.join_device(R, S, 0); torch.cuda.synchronize(); first = (time.perf_counter() - t0) * 1e3
ts = []
for i in range(10):
    ex.join_device(R, S, 0); ts.append(ex.last_timing())
m = lambda k: sum(t[k] for t in ts[2:]) / len(ts[2:])
a = m("ms_scatter_pass0") / 2
print("HMJ_PLACE=%s first join %.0f ms | steady total %.3f passA %.3f passB(build) %.3f passB(probe) %.3f probe %.3f" % (
    os.environ.get("HMJ_PLACE", "1"), first, m("ms_total"), a, m("ms_partition_build") - a, m("ms_partition_probe") - a, m("ms_probe_count")), flush=True)
