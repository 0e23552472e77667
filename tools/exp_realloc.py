#!/usr/bin/env python3
"""Placement lottery inside ONE process: a fresh Executor per round (its buffers are freed at close), dummy tensors of
varying size in between.  With HMJ_PLACE=1 HMJ_TRACE=1 every round prints what the first join measured for its
freshly allocated slab buffers (stderr) next to the steady-state times of the round."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import hashmergejoin_amd as H

n = 1 << 28
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 8
gen = H.Executor(0)
R, S = gen.gen_build(n), gen.gen_probe(n, n)
dummies = []
for rnd in range(rounds):
    ex = H.Executor(0)
    ex.set_profiling(True)
    ts = []
    for i in range(6):
        r = ex.join_device(R, S, 0)
        t = ex.last_timing()
        if i >= 2:
            ts.append(t)
    m = lambda k: sum(t[k] for t in ts) / len(ts)
    a = m("ms_scatter_pass0") / 2
    sys.stderr.write("== steady round %d: passA %.3f  passB(build) %.3f  passB(probe) %.3f  total %.3f probe %.3f\n" % (
        rnd, a, m("ms_partition_build") - a, m("ms_partition_probe") - a, m("ms_total"), m("ms_probe_count")))
    sys.stderr.flush()
    ex.close()
    dummies.append(torch.empty(((rnd * 7) % 5 + 1) * 211 * (1 << 20), dtype=torch.uint8, device="cuda"))
    if rnd % 3 == 2:
        dummies.pop(0)
