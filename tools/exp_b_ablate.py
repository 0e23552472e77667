#!/usr/bin/env python3
"""Timing-only ablations of slab pass B's gather, alternated inside ONE process (same buffers: the placement
lottery cancels).  Needs a -DHMJ_DEV build: HMJ_LIB=build/variants/libhmj_base.so python tools/exp_b_ablate.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401
import hashmergejoin_amd as H

log2n = int(sys.argv[1]) if len(sys.argv) > 1 else 28
ex = H.Executor(0)
ex.set_profiling(True)
n = 1 << log2n
R, S = ex.gen_build(n), ex.gen_probe(n, n)
L = ex.L
for i in range(3):
    ex.join_device(R, S, 0)
modes = [(0, 0, "gather (default lookup)"), (1, 0, "gather (other lookup)"), (0, 2, "128-B aligned gather"),
         (0, 1, "contiguous read + plan"), (1, 1, "contiguous + other plan"), (0, 4, "contiguous, no plan")]
res = {m[:2]: [] for m in modes}
# correctness of the other lookup first: same join, same sums
ref = ex.join_device(R, S, 0)
L.hmj_dev_set_b_addr_alt(1)
alt = ex.join_device(R, S, 0)
L.hmj_dev_set_b_addr_alt(0)
assert (int(ref.n_matches), int(ref.sum_r), int(ref.sum_s)) == (int(alt.n_matches), int(alt.sum_r), int(alt.sum_s)) and int(alt.n_matches) == n
assert ex.last_timing()["path"] & 1
# hmj_prepare_build partitions the build side only (slab A + B into slab_br, no probe kernel that would notice
# the wrong rows), so every mode runs to the end on the same buffers
for rep in range(6):
    for alt_on, abl, _ in modes:
        assert L.hmj_dev_set_b_ablate(abl) == 0
        L.hmj_dev_set_b_addr_alt(alt_on)
        ex.prepare_build(R, n)
        t = ex.last_timing()
        res[(alt_on, abl)].append((t["ms_scatter_pass0"], t["ms_scatter_pass1"], t["path"]))
L.hmj_dev_set_b_ablate(0)
L.hmj_dev_set_b_addr_alt(0)
ex.prepare_build(R, n)  # (replaces the prepared build side the last ablation left behind)
r = ex.join_device(R, S, 0)
assert int(r.n_matches) == n
for alt_on, abl, name in modes:
    x = res[(alt_on, abl)]
    print("%-26s passA %.3f | passB %s | mean %.3f | path %s" % (
        name, sum(v[0] for v in x) / len(x), " ".join("%.3f" % v[1] for v in x), sum(v[1] for v in x) / len(x),
        sorted(set(hex(v[2]) for v in x))), flush=True)
